// Persistent few-row decode layers for gfx950 (MI355X): ALL decoder layers of one decode step for 1 - 4 rows in ONE launch.
//
// What it replaces: the reference runs one HF-eager LlamaDecoderLayer per branch and token (experiments/llava/model/language_model/
// llava_llama.py:88-103 -> transformers LlamaDecoderLayer [ext]; one question in flight = 2 - 3 branch rows, llava_calibrate.py:130,
// 161-177).  The five-launch layer of vdd_llm_kernels.hip (normed qkv, fused attention, o + residual, normed gate/up + SwiGLU,
// down + residual) spends ~30 of its ~95 us per layer in the ramp-up / drain of five weight-streaming launches; here the weight
// stream of a CU never stops:
//
//   grid = G workgroups (one per CU, G = min(#CUs, d / 16)), 8 waves each:
//     waves 0-3  STREAM: each owns a quarter of K; W rows go global -> registers (MFMA 16x16x32 B fragments, 16 B per lane) through a
//                THREE-batch register pipeline (3 x 8 fragments = 24 KiB per wave, 96 KiB per CU in flight) that runs ahead across
//                column blocks AND across the op boundaries o -> gate/up -> down -> next layer's qkv (the prefetch credit: while a CU
//                waits for an activation vector its next 96 KiB of weights are already on their way);
//     waves 4-7  GATHER: bring every op's input vector on chip.  An op's output leaves its producer as 8-byte {tag, 2 elements}
//                granules (ONE write-through sc1 store each: the data is the flag); a gather wave sweeps the granules with sc1 loads
//                until every tag carries the phase's epoch, writes the values into the LDS image the MFMA A fragments are read from,
//                and (for the two normalised inputs) sums the squares, normalises in place (rmsnorm_kernel's roundings) and raises
//                an LDS counter the stream waves wait on.  No grid barrier, no fences, no polling by the stream waves.
//   per layer:   [G1] h -> rmsnorm(ln1) -> X   qkv  -> granules QKV
//                [G2] q,k,v of (row, head) -> RoPE -> LDS; KV-cache write     attention of item (head, key slice) over all rows
//                                                                              -> granules PART (un-normalised partials)
//                [G3] merge the slices of unit (row, head) -> granules AO
//                [G4] AO -> X            o-proj + residual -> h'  -> granules HP  (the residual of a CU's 16 columns never leaves it)
//                [G5] h' -> rmsnorm(ln2) -> X   gate/up + SwiGLU -> granules ACT
//                [G6] ACT -> X_F         down + residual -> h'' -> granules HPP (last layer: plain [M, d] + per-block sums of squares)
//   Every spin is bounded (SPIN_LIMIT ticks of the 100 MHz clock): a wave that gives up raises an LDS abort word that short-cuts
//   every later wait of its workgroup, and the code lands in ctrl[1] (the host raises).  Epochs: tag = 1 + 1024 * launch + 8 * layer
//   + buffer, with the launch count kept in device memory (ctrl[0]) so that graph replays get fresh tags without a memset node.
//
// Rounding points are those of the five-launch layer (bf16 / fp16 after every torch op of the reference); accumulation order
// differs (K in four contiguous quarters, key slices per CU), so results agree to accumulation noise, not bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdd_elem.h"
#include "vdd_lost.h"

namespace {
namespace VDD_ELEM_NS {
using namespace vdd_elem;

typedef __attribute__((ext_vector_type(8))) short frag8_t;
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) const frag8_t* g_frag_p;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(1))) const u32x4_t* g_u4_p;
typedef __attribute__((address_space(1))) u64* g_u64_p;
typedef __attribute__((address_space(1))) unsigned* g_u32_p;
typedef __attribute__((address_space(1))) const uint16_t* g_u16_p;
typedef __attribute__((address_space(1))) const uint32_t* g_cu32_p;
// LDS: explicit address space everywhere - one generic (flat) access in a stream wave makes the compiler wait for vmcnt(0) AND
// lgkmcnt(0) at every later use (a flat access may hit either), which drains the weight pipeline
#define LDS(T) __attribute__((address_space(3))) T
typedef LDS(uint32_t) lds_u32;
typedef LDS(float) lds_f32;
typedef LDS(f32x4_t) lds_f32x4;
typedef LDS(u32x4_t) lds_u32x4;
typedef LDS(frag8_t) lds_frag;

struct AttnRow { int slot, len, pslot, plen; };

struct Params {
    const vdd_layer_desc* layers;
    int n_layers, M, d, H, F, S, G, has_bias;
    float eps, scale;
    const uint16_t* resid_in;
    uint16_t* resid_out;
    float* ss_out;
    const int* pos; const int* cpos; const int* slot;
    const float* cs;
    const AttnRow* rows;
    long long slot_stride, pre_stride;
    int t_max, pre_tmax;
    unsigned* ctrl;                                   // [0] launch count, [1] first give-up code
    u64 *g_qkv, *g_part, *g_ao, *g_hp, *g_act, *g_hpp;
    u64* dbg;                                         // optional [G][n_layers][16] timeline (100 MHz ticks), or nullptr
};

#ifndef VDD_PL_STAGES
#define VDD_PL_STAGES 3                                // weight batches in flight per stream wave (8 KiB each)
#endif
#ifndef VDD_PL_POLL_SLEEP
#define VDD_PL_POLL_SLEEP 2                            // s_sleep between two sweeps of a chunk whose granules are not all there
#endif
constexpr long long SPIN_LIMIT = 5000000;             // 50 ms of the 100 MHz realtime clock
constexpr int PART_G = 132;                           // granules per attention partial: 128 sums, max, weight, 2 pad
enum { F_XREADY = 0, F_SDONE, F_QKVREADY, F_ATTARR, F_GBAR, F_ABORT, F_ARRIVE = 8, F_FREED = 12, F_N = 16 };
enum { T_QKV = 0, T_PART, T_AO, T_HP, T_ACT, T_HPP };

struct Lds { uint32_t xd, xf, part, qkvs, attp, resid, bias, gred, flags, rowi, desc, total; };
constexpr int DESC_N = 11;                           // pointers per vdd_layer_desc
__host__ __device__ inline Lds lds_layout(int M, int d, int F, int rmax, int n_layers) {
    Lds L; uint32_t o = 0;
    L.xd = o;    o += (uint32_t)M * 2u * (uint32_t)d;
    L.xf = o;    o += (uint32_t)M * 2u * (uint32_t)F;
    L.part = o;  o += 4u * 4u * 16u * 16u;                      // [buffer][wave][column] float4 (rows 0..3)
    L.qkvs = o;  o += (uint32_t)rmax * 3u * 256u;               // [row][q|k|v][128] elements, q and k rotated
    L.attp = o;  o += 4u * (uint32_t)rmax * PART_G * 4u;        // [wave][row][132] floats
    L.resid = o; o += 2u * (uint32_t)rmax * 16u * 4u;           // [column block ordinal][row][column] floats (rounded values)
    L.bias = o;  o += 4u * 16u * 4u;                            // qkv bias of this CU's (<= 4) column blocks
    L.gred = o;  o += 2u * 4u * 4u * 4u;                        // [parity][gather wave][row]
    L.flags = o; o += F_N * 4u;
    L.rowi = o;  o += 4u * 32u + 16u;                                // per row: key slice of this CU, pool offsets (RowInfo)
    L.desc = o;  o += (uint32_t)n_layers * DESC_N * 8u;        // the layer descriptors: the stream waves must not touch global memory for them
    L.total = (o + 15u) & ~15u;
    return L;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
    float s = dot2(a.x, b.x, 0.f);
    s = dot2(a.y, b.y, s);
    s = dot2(a.z, b.z, s);
    s = dot2(a.w, b.w, s);
    return s;
}
__device__ __forceinline__ uint32_t norm_pair(uint32_t hv, uint32_t gv, float rstd) {       // rmsnorm_kernel's roundings
    const uint32_t nb = cvt_pk(lo(hv) * rstd, hi(hv) * rstd);
    return cvt_pk(lo(nb) * lo(gv), hi(nb) * hi(gv));
}
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

// One granule = one aligned 8-byte write-through store.  Issued from asm so that the compiler's s_waitcnt bookkeeping of the stream
// waves sees loads only: with a store pending beside loads it waits for vmcnt(0) at the next fragment use (it must assume the two
// kinds retire out of order), which would drain the four-batch weight pipeline once per column block.  An uncounted store can only
// make a later vmcnt(N) wait longer, never shorter (loads retire in order among themselves).
__device__ __forceinline__ void st_granule(u64* p, unsigned tag, unsigned val) {
    const u64 x = ((u64)tag << 32) | val;
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(x) : "memory");
}
// plain stores of the stream waves (the step's output), uncounted for the same reason
__device__ __forceinline__ void st_u16_asm(uint16_t* p, uint32_t v) { asm volatile("global_store_short %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_f32_asm(float* p, float v) { asm volatile("global_store_dword %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory"); }
// timeline stamp (debug launches only): slot k of (workgroup, layer)
__device__ __forceinline__ void stamp(u64* dbg, int cu, int NL, int layer, int k, int lane) {
    if (dbg != nullptr && lane == 0) {
        const u64 t = __builtin_amdgcn_s_memrealtime();
        u64* q = dbg + ((size_t)cu * NL + layer) * 32 + k;
        asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(q), "v"(t) : "memory");
    }
}
__device__ __forceinline__ u64 ld_granule(const u64* p) {
    return __hip_atomic_load((g_u64_p)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- LDS words shared by the waves of a workgroup (monotonic counters; LDS executes a wave's operations in order)
__device__ __forceinline__ uint32_t flag_ld(lds_u32* f, int i) { return ((volatile lds_u32*)f)[i]; }
__device__ __forceinline__ void flag_add(lds_u32* f, int i) {
    asm volatile("" ::: "memory");
    __hip_atomic_fetch_add(f + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// wait until counter i >= target.  No vector-memory instruction in here (see st_granule): a give-up only raises the LDS abort word.
__device__ __forceinline__ bool flag_wait(lds_u32* f, int i, uint32_t target, uint32_t code) {
    volatile lds_u32* vf = f;
    asm volatile("" ::: "memory");
    if (vf[i] >= target) { asm volatile("" ::: "memory"); return true; }
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 1;; ++it) {
        __builtin_amdgcn_s_sleep(1);
        if (vf[i] >= target) { asm volatile("" ::: "memory"); return true; }
        if (vf[F_ABORT] != 0) return false;
        if ((it & 127u) == 0 && (long long)__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) { vf[F_ABORT] = code; return false; }
    }
}

// ======================================================================================================================
template <int RMAX>
__global__ void __launch_bounds__(512) decode_layers_kernel(const Params p) {
    static_assert(RMAX == 2 || RMAX == 4, "row buckets");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, d = p.d, H = p.H, F = p.F, S = p.S, G = p.G, NL = p.n_layers;
    const Lds L = lds_layout(M, d, F, RMAX, NL);
    lds_u32* const flags = (lds_u32*)(smem + L.flags);
    const int cu = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, g = lane >> 4;

    // ---- workgroup start: counters to zero, the residual of this CU's d-wide column blocks (layer 0: the embeddings)
    if (tid < F_N) flags[tid] = 0;
    if (tid < 2 * RMAX * 16) {
        const int i = tid / (RMAX * 16), r = (tid / 16) % RMAX, c = tid & 15;
        float v = 0.f;
        if (i * 16 + c < d / G && r < M) v = e2f(((g_u16_p)p.resid_in)[(size_t)r * d + cu * (d / G) + i * 16 + c]);
        ((lds_f32*)(smem + L.resid))[(i * RMAX + r) * 16 + c] = v;
    }
    if (tid == 64 && cu < H * S) {
        // the rows' key ranges and pool offsets (the same for every layer): this CU's attention item is (head, key slice sl) over ALL
        // rows.  An LDS table, read back per round with a run-time row (a select chain over register arrays becomes scratch indexing)
        const int head = cu / S, sl = cu % S;
        int n_rounds = 0;
        lds_u32* t = (lds_u32*)(smem + L.rowi);
        for (int r = 0; r < RMAX; ++r) {
            const AttnRow ar = p.rows[r < M ? r : M - 1];
            const int n_all = ar.len - 1;                           // keys in the cache; the new token is attended from LDS
            const int per = (((n_all + S - 1) / S) + 15) & ~15;
            const int k_lo = min(sl * per, n_all), k_hi = min(k_lo + per, n_all);
            const long long own_off = (long long)ar.slot * p.slot_stride + (long long)head * p.t_max * 128 - (long long)ar.plen * 128;
            const long long pre_off = (long long)ar.pslot * p.pre_stride + (long long)head * p.pre_tmax * 128;
            t[r * 8 + 0] = (uint32_t)k_lo; t[r * 8 + 1] = (uint32_t)k_hi; t[r * 8 + 2] = (uint32_t)ar.plen; t[r * 8 + 3] = (uint32_t)n_rounds;
            t[r * 8 + 4] = (uint32_t)(u64)own_off; t[r * 8 + 5] = (uint32_t)((u64)own_off >> 32);
            t[r * 8 + 6] = (uint32_t)(u64)pre_off; t[r * 8 + 7] = (uint32_t)((u64)pre_off >> 32);
            if (r < M) n_rounds += (k_hi - k_lo + 15) >> 4;
        }
        t[32] = (uint32_t)n_rounds;
    }
    for (int i = tid; i < NL * DESC_N; i += 512)
        ((LDS(u64)*)(smem + L.desc))[i] = ((__attribute__((address_space(1))) const u64*)p.layers)[i];
    const unsigned launch = __hip_atomic_load((g_u32_p)p.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tag0 = 1u + launch * 1024u;
    __syncthreads();
    // pointer `field` of layer `layer` as a wave-uniform value (one broadcast LDS read: a global load here would be a VECTOR load
    // - the compiler cannot prove the table unclobbered - queued behind the weight pipeline, and waited for with vmcnt(0))
    auto desc = [&](int layer, int field) __attribute__((always_inline)) {
        const u64 v = ((LDS(u64)*)(smem + L.desc))[layer * DESC_N + field];
        const uint32_t lo32 = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const uint16_t*)(((u64)hi32 << 32) | lo32);
    };
    enum { D_LN1 = 0, D_WQKV, D_BQKV, D_WO, D_LN2, D_WGU, D_WD, D_KOWN, D_VOWN, D_KPRE, D_VPRE };
    // gate/up features of this CU: F / G each; when that is odd, an even CU takes one more and its odd neighbour one less, so that
    // every range starts on an even feature (a granule carries the PAIR (f, f + 1) and is written by one store)
    const int fpc = F / G;
    const int gu_n = (fpc & 1) ? ((cu & 1) ? fpc - 1 : fpc + 1) : fpc;
    const int gu_base = cu * fpc + ((fpc & 1) && (cu & 1) ? 1 : 0);

    if (wave < 4) {
        // =============================================================================================== STREAM waves
        const int rr = ln < M ? ln : M - 1;                   // A-fragment row of this lane (rows >= M repeat the last one)
        const uint32_t xsw = (uint32_t)(rr & 15) << 4;
        struct Cur { int layer, op, i, b, nb, ncb, kq, K, npc; bool valid; const uint16_t* W; const uint16_t* w; };
        auto op_setup = [&](Cur& c) __attribute__((always_inline)) {
            c.K = c.op == 3 ? F : d;
            c.kq = c.K >> 2;
            c.nb = (c.kq + 255) >> 8;
            c.npc = c.op == 0 ? 3 * d / G : (c.op == 2 ? gu_n : d / G);       // this CU's columns (gate/up: features) of the op
            c.ncb = c.op == 2 ? (c.npc + 7) >> 3 : (c.npc + 15) >> 4;           // in blocks of 16 MFMA columns, the last one ragged
            c.W = desc(c.layer, c.op == 0 ? D_WQKV : c.op == 1 ? D_WO : c.op == 2 ? D_WGU : D_WD);
        };
        auto cb_setup = [&](Cur& c) __attribute__((always_inline)) {
            // Every CU owns the SAME number of contiguous columns of every op (N / G: equal HBM bytes per CU - with whole 16-column blocks
            // dealt round the CUs, 96 of 256 CUs streamed a sixth block of gate/up and everybody waited for them at the next hand-off).
            // A ragged last block reads the CU's last valid row again (an L1 hit, no HBM traffic) and stores nothing for it.
            size_t row;
            if (c.op == 2) { int fi = c.i * 8 + (ln & 7); if (fi >= c.npc) fi = c.npc - 1; row = (size_t)((ln < 8 ? 0 : F) + gu_base + fi); }
            else { int ci = c.i * 16 + ln; if (ci >= c.npc) ci = c.npc - 1; row = (size_t)(cu * c.npc + ci); }
            c.w = c.W + row * (size_t)c.K + (size_t)wave * c.kq + g * 8;
        };
        // next batch of this wave's stream.  A segment ends in front of the attention output projection (op 1: the attention runs
        // first and re-validates the cursor) and behind the last layer's down projection.
        auto advance = [&](Cur& c) __attribute__((always_inline)) {
            if (++c.b < c.nb) return;
            c.b = 0;
            if (++c.i < c.ncb) { cb_setup(c); return; }
            c.i = 0;
            if (++c.op == 4) { c.op = 0; ++c.layer; }
            if (c.op == 1 || (c.op == 0 && c.layer == NL)) { c.valid = false; return; }
            op_setup(c); cb_setup(c);
        };
        const uint16_t* const dummy = (const uint16_t*)p.ctrl;   // a valid 16-byte-aligned line for the loads of a finished stream
        auto ldw = [&](frag8_t (&w)[8], const Cur& c) __attribute__((always_inline)) {
            const uint16_t* base = c.valid ? c.w : dummy;
            const int kk = c.valid ? c.b << 8 : 0, kq = c.valid ? c.kq : 32;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int k = kk + 32 * u;
                if (k >= kq) k = kq - 32;                      // ragged last batch: a valid address, multiplied by a zero fragment
                w[u] = *(g_frag_p)(base + k);
            }
        };
        auto mm = [&](const frag8_t (&w)[8], const Cur& c, f32x4_t& acc) __attribute__((always_inline)) {
            const uint32_t xb = (c.op == 3 ? L.xf : L.xd) + (uint32_t)rr * 2u * (uint32_t)c.K;
            const int kk = c.b << 8;
            const frag8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int k = kk + 32 * u;
                const bool in = k < c.kq;
                if (!in) k = c.kq - 32;
                const uint32_t off = ((uint32_t)(wave * c.kq + k + g * 8) * 2u) ^ xsw;
                const frag8_t a = *(lds_frag*)(smem + xb + off);
                acc = mfma16(a, in ? w[u] : zero, acc);
            }
        };
        lds_f32* const resid = (lds_f32*)(smem + L.resid);
        lds_f32* const bias_s = (lds_f32*)(smem + L.bias);
        lds_f32x4* const part = (lds_f32x4*)(smem + L.part);
        uint32_t cbseq = 0;

        // ---- a column block is complete in this wave: partial sums to LDS; wave (seq & 3) adds the four and runs the epilogue
        auto colblock_end = [&](const Cur& c, const f32x4_t& acc) __attribute__((always_inline)) {
            const uint32_t seq = cbseq++;
            const int buf = seq & 3;
            flag_wait(flags, F_FREED + buf, seq >> 2, 0x100u | (uint32_t)c.op);
            if (g == 0) part[(buf * 4 + wave) * 16 + ln] = acc;
            if (lane == 0) {
                flag_add(flags, F_ARRIVE + buf);
                if (c.i == c.ncb - 1) flag_add(flags, F_SDONE);       // this wave has read its last X fragment of the op
            }
            if (buf != wave) return;
            if (c.op == 3 && c.i == c.ncb - 1) stamp(p.dbg, cu, NL, c.layer, 22, lane);
            flag_wait(flags, F_ARRIVE + buf, 4u * ((seq >> 2) + 1u), 0x200u | (uint32_t)c.op);
            if (c.op == 3 && c.i == c.ncb - 1) stamp(p.dbg, cu, NL, c.layer, 23, lane);
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            if (g == 0) {
                const f32x4_t p0 = part[(buf * 4 + 0) * 16 + ln], p1 = part[(buf * 4 + 1) * 16 + ln], p2 = part[(buf * 4 + 2) * 16 + ln],
                               p3 = part[(buf * 4 + 3) * 16 + ln];
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] = (p0[r] + p1[r]) + (p2[r] + p3[r]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_add(flags, F_FREED + buf);
            const int col = cu * c.npc + c.i * 16 + ln;                // output column of this lane (gate/up: see below)
            const bool cval = c.i * 16 + ln < c.npc;
            const unsigned tagL = tag0 + (unsigned)c.layer * 8u;
            const bool last = c.layer == NL - 1;
            if (c.op == 0) {                                          // qkv: (+ bias) -> granules QKV[row][column pair]
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    float o = rnd(s[r]);
                    if (p.has_bias) o = rnd(o + bias_s[c.i * 16 + ln]);
                    const uint32_t ob = f2e(o);
                    const uint32_t other = __shfl_xor(ob, 1);
                    if (g == 0 && (ln & 1) == 0 && r < M && cval)
                        st_granule(p.g_qkv + (size_t)r * (3 * d / 2) + col / 2, tagL + T_QKV, ob | (other << 16));
                }
            } else if (c.op == 2) {                                   // gate (columns 0-7) | up (8-15) of 8 features: SwiGLU
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    const float up = __shfl_down(s[r], 8);
                    const float gb = rnd(s[r]), ub = rnd(up);
                    const float sl = rnd(gb / (1.f + __expf(-gb)));
                    const uint32_t ab = f2e(sl * ub);
                    const uint32_t other = __shfl_xor(ab, 1);
                    if (g == 0 && ln < 8 && (ln & 1) == 0 && r < M && c.i * 8 + ln < c.npc)
                        st_granule(p.g_act + (size_t)r * (F / 2) + (gu_base + c.i * 8 + ln) / 2, tagL + T_ACT, ab | (other << 16));
                }
            } else {                                                  // o / down: + residual = the new residual stream
                u64* const gout = c.op == 1 ? p.g_hp : p.g_hpp;
                const unsigned tg = tagL + (c.op == 1 ? T_HP : T_HPP);
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    float h = 0.f;
                    if (g == 0 && r < M && cval) {
                        h = rnd(rnd(s[r]) + resid[(c.i * RMAX + r) * 16 + ln]);
                        resid[(c.i * RMAX + r) * 16 + ln] = h;
                    }
                    const uint32_t hb = f2e(h);
                    const uint32_t other = __shfl_xor(hb, 1);
                    if (c.op == 3 && last) {                          // the step's output: plain [M, d] + sums of squares per block
                        float sq = h * h;
                        sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4); sq += __shfl_xor(sq, 8);
                        if (g == 0 && r < M) {
                            if (cval) st_u16_asm(p.resid_out + (size_t)r * d + col, hb);
                            if (ln == 0) st_f32_asm(p.ss_out + (size_t)r * (G * c.ncb) + cu * c.ncb + c.i, sq);
                        }
                    } else if (g == 0 && (ln & 1) == 0 && r < M && cval) {
                        st_granule(gout + (size_t)r * (d / 2) + col / 2, tg, hb | (other << 16));
                    }
                }
                if (c.op == 3 && c.i == c.ncb - 1) stamp(p.dbg, cu, NL, c.layer, 24, lane);
            }
        };

        struct RowInfo { int k_lo, k_hi, plen, rstart; long long own_off, pre_off; };
        const int sl = cu % S, head = cu / S;
        int rstart[RMAX];                                           // rounds of 16 keys, the rows' slices end to end (table: prologue)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) rstart[r] = (int)((lds_u32*)(smem + L.rowi))[r * 8 + 3];
        const int n_rounds = (int)((lds_u32*)(smem + L.rowi))[32];
        frag8_t w0[8], w1[8], w2[8];
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
        Cur C, P;
        C.layer = 0; C.op = 0; C.i = 0; C.b = 0; C.valid = true;
        op_setup(C); cb_setup(C);
        P = C;
        // batches of this wave per (op, CU): the same for every layer
        auto op_batches = [&](int op) __attribute__((always_inline)) {
            const int K = op == 3 ? F : d, nb = ((K >> 2) + 255) >> 8;
            const int npc = op == 0 ? 3 * d / G : (op == 2 ? gu_n : d / G);
            return (op == 2 ? (npc + 7) >> 3 : (npc + 15) >> 4) * nb;
        };
        const int nb_qkv = op_batches(0), nb_rest = op_batches(1) + op_batches(2) + op_batches(3);

        // One batch: 8 MFMAs on the stage's fragments, then the stage is refilled with the batch THREE ahead.  Every weight load of
        // the loop is issued UNCONDITIONALLY (a stream that has reached the end of its segment reads one dummy line): hipcc places
        // s_waitcnt statically, and a load that sits behind a condition makes it assume at every fragment use that the batches
        // issued after it may not exist - i.e. it waits for all of them (vmcnt(7..0) instead of vmcnt(23..16): a one-batch pipeline).
#define VDD_STEP(WS)                                                                                                   \
    {                                                                                                                  \
        if (C.b == 0) {                                                                                                \
            if (C.i == 0) {                                                                                            \
                if (wave == 0) stamp(p.dbg, cu, NL, C.layer, C.op * 2, lane);                                          \
                flag_wait(flags, F_XREADY, 4u * (uint32_t)(4 * C.layer + C.op + 1), 0x300u | (uint32_t)C.op);          \
                if (wave == 0) stamp(p.dbg, cu, NL, C.layer, C.op * 2 + 1, lane);                                      \
            }                                                                                                          \
            acc = f32x4_t{0.f, 0.f, 0.f, 0.f};                                                                         \
        }                                                                                                              \
        mm(WS, C, acc);                                                                                                \
        ldw(WS, P);                                                                                                    \
        if (P.valid) advance(P);                                                                                       \
        if (C.b == C.nb - 1) colblock_end(C, acc);                                                                     \
        advance(C);                                                                                                    \
    }

        for (;;) {
            // the segment's batches: qkv of layer 0, then per layer [o, gate/up, down, next layer's qkv]
            int rem = C.op == 0 ? nb_qkv : nb_rest + (C.layer + 1 < NL ? nb_qkv : 0);
            ldw(w0, P); if (P.valid) advance(P);
            ldw(w1, P); if (P.valid) advance(P);
#if VDD_PL_STAGES == 3
            ldw(w2, P); if (P.valid) advance(P);
            for (; rem >= 3; rem -= 3) {
                VDD_STEP(w0)
                VDD_STEP(w1)
                VDD_STEP(w2)
            }
            if (rem >= 1) VDD_STEP(w0)
            if (rem >= 2) VDD_STEP(w1)
#else
            for (; rem >= 2; rem -= 2) {
                VDD_STEP(w0)
                VDD_STEP(w1)
            }
            if (rem >= 1) VDD_STEP(w0)
#endif
            if (C.layer >= NL) break;
            // the stages hold dummy lines now: tell the register allocator (an empty asm that DEFINES them), so that the attention's
            // K / V batches can live in their registers
#pragma unroll
            for (int u = 0; u < 8; ++u) { asm volatile("" : "=v"(w0[u])); asm volatile("" : "=v"(w1[u])); asm volatile("" : "=v"(w2[u])); }
            // ------------------------------------------------------------------------------ attention of layer C.layer
            const int Lc = C.layer;
            if (cu < H * S) {
                // Work = ROUNDS of 16 keys (one key per 16-lane group), the rows' key slices laid end to end: NR rounds are in flight at a
                // time whatever the rows' lengths (a main branch with 80 keys per slice and an image-free branch with 10 share one
                // 8-round batch), and the first batch is issued BEFORE the rotated query exists (its addresses depend on neither).
                constexpr int NR = 8;
                const int grp = wave * 4 + g;                       // 16 key groups per CU, 16 lanes x 8 dims each
                const uint16_t* const ko = desc(Lc, D_KOWN); const uint16_t* const vo = desc(Lc, D_VOWN);
                const uint16_t* const kp = desc(Lc, D_KPRE); const uint16_t* const vp = desc(Lc, D_VPRE);
                uint4 kf[NR], vf[NR];
                auto locate = [&](int q, int& row, int& j) __attribute__((always_inline)) {     // round q -> (row, round of the row)
                    row = 0;
#pragma unroll
                    for (int r = 1; r < RMAX; ++r) if (q >= rstart[r]) row = r;
                    int st = 0;
#pragma unroll
                    for (int r = 1; r < RMAX; ++r) st = row == r ? rstart[r] : st;
                    j = q - st;
                };
                auto rowinfo = [&](int row) __attribute__((always_inline)) {
                    const u32x4_t x = *(lds_u32x4*)(smem + L.rowi + row * 32), y = *(lds_u32x4*)(smem + L.rowi + row * 32 + 16);
                    RowInfo ri;
                    ri.k_lo = (int)x[0]; ri.k_hi = (int)x[1]; ri.plen = (int)x[2]; ri.rstart = (int)x[3];
                    ri.own_off = (long long)(((u64)y[1] << 32) | y[0]); ri.pre_off = (long long)(((u64)y[3] << 32) | y[2]);
                    return ri;
                };
                auto fetch = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        int row, j; locate(q0 + u < n_rounds ? q0 + u : (n_rounds > 0 ? n_rounds - 1 : 0), row, j);
                        const RowInfo ri = rowinfo(row);
                        const int t = ri.k_lo + 16 * j + grp;
                        const int tt = t < ri.k_hi ? t : (ri.k_hi > 0 ? ri.k_hi - 1 : 0);
                        const bool pre = tt < ri.plen;
                        const long long eo = (pre ? ri.pre_off : ri.own_off) + (long long)tt * 128 + ln * 8;     // element offset
                        kf[u] = __builtin_bit_cast(uint4, *(g_u4_p)((pre ? kp : ko) + eo));
                        vf[u] = __builtin_bit_cast(uint4, *(g_u4_p)((pre ? vp : vo) + eo));
                    }
                };
                if (wave == 0) stamp(p.dbg, cu, NL, Lc, 8, lane);
                fetch(0);
                flag_wait(flags, F_QKVREADY, 4u * (uint32_t)(Lc + 1), 0x400u);
                if (wave == 0) stamp(p.dbg, cu, NL, Lc, 9, lane);
                lds_f32* const attp = (lds_f32*)(smem + L.attp);
                float m[RMAX], l[RMAX], a[RMAX][8];
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    m[r] = -INFINITY; l[r] = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[r][e] = 0.f;
                }
                for (int q0 = 0; q0 < n_rounds; q0 += NR) {
                    if (q0 != 0) fetch(q0);                         // more than 8 rounds per CU (contexts beyond ~1,000 keys): one more round trip
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        if (q0 + u < n_rounds) {
                            int row, j; locate(q0 + u, row, j);
                            const uint4 qr = __builtin_bit_cast(uint4, *(lds_u32x4*)(smem + L.qkvs + row * 768 + ln * 16));   // the row's rotated query
                            float sc = dot8(qr, kf[u]);
                            sc += __shfl_xor(sc, 1); sc += __shfl_xor(sc, 2); sc += __shfl_xor(sc, 4); sc += __shfl_xor(sc, 8);
                            sc *= p.scale;
                            const RowInfo ri = rowinfo(row);
                            const bool live = ri.k_lo + 16 * j + grp < ri.k_hi;
#pragma unroll
                            for (int r = 0; r < RMAX; ++r)
                                if (row == r && live) ATT_ONLINE_STEP(sc, vf[u], m[r], l[r], a[r]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    if (r >= M) continue;
                    const unsigned char* qs = smem + L.qkvs + r * 768;
                    if (sl == S - 1) {                              // the new token (rotated k, v staged by the gather wave)
                        const uint4 kn = __builtin_bit_cast(uint4, *(lds_u32x4*)(qs + 256 + ln * 16));
                        const uint4 vn = __builtin_bit_cast(uint4, *(lds_u32x4*)(qs + 512 + ln * 16));
                        const uint4 qr = __builtin_bit_cast(uint4, *(lds_u32x4*)(qs + ln * 16));
                        float sc = dot8(qr, kn);
                        sc += __shfl_xor(sc, 1); sc += __shfl_xor(sc, 2); sc += __shfl_xor(sc, 4); sc += __shfl_xor(sc, 8);
                        sc *= p.scale;
                        if (wave == 3 && g == 3) ATT_ONLINE_STEP(sc, vn, m[r], l[r], a[r]);
                    }
#pragma unroll
                    for (int o = 16; o <= 32; o <<= 1) {
                        const float mo = __shfl_xor(m[r], o), lo_ = __shfl_xor(l[r], o);
                        const float mn = fmaxf(m[r], mo);
                        const float e0 = (m[r] == -INFINITY) ? 0.f : __expf(m[r] - mn), e1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
                        l[r] = l[r] * e0 + lo_ * e1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float ao = __shfl_xor(a[r][e], o); a[r][e] = a[r][e] * e0 + ao * e1; }
                        m[r] = mn;
                    }
                    if (g == 0) {
                        lds_f32* pw = attp + (wave * RMAX + r) * PART_G;
                        *(lds_f32x4*)(pw + ln * 8) = f32x4_t{a[r][0], a[r][1], a[r][2], a[r][3]};
                        *(lds_f32x4*)(pw + ln * 8 + 4) = f32x4_t{a[r][4], a[r][5], a[r][6], a[r][7]};
                        if (ln == 0) { pw[128] = m[r]; pw[129] = l[r]; }
                    }
                }
                if (lane == 0) flag_add(flags, F_ATTARR);
                if (wave == 0) {                                    // the CU's partial of (row, head, slice): four waves merged
                    flag_wait(flags, F_ATTARR, 4u * (uint32_t)(Lc + 1), 0x500u);
#pragma unroll
                    for (int r = 0; r < RMAX; ++r) {
                        if (r >= M) continue;
                        float Mx = -INFINITY;
#pragma unroll
                        for (int w = 0; w < 4; ++w) Mx = fmaxf(Mx, attp[(w * RMAX + r) * PART_G + 128]);
                        float Ls = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            lds_f32* pw = attp + (w * RMAX + r) * PART_G;
                            const float wgt = pw[128] == -INFINITY ? 0.f : __expf(pw[128] - Mx);
                            Ls += wgt * pw[129];
                            a0 += wgt * pw[2 * lane]; a1 += wgt * pw[2 * lane + 1];
                        }
                        u64* gp = p.g_part + ((size_t)(r * H + head) * S + sl) * PART_G;
                        const unsigned tg = tag0 + (unsigned)Lc * 8u + T_PART;
                        st_granule(gp + 2 * lane, tg, __builtin_bit_cast(unsigned, a0));
                        st_granule(gp + 2 * lane + 1, tg, __builtin_bit_cast(unsigned, a1));
                        if (lane == 0) {
                            st_granule(gp + 128, tg, __builtin_bit_cast(unsigned, Mx));
                            st_granule(gp + 129, tg, __builtin_bit_cast(unsigned, Ls));
                        }
                    }
                }
            }
            if (wave == 0) stamp(p.dbg, cu, NL, Lc, 10, lane);
            C.valid = true;
            op_setup(C); cb_setup(C);
            P = C;
        }
#undef VDD_STEP
    } else {
        // =============================================================================================== GATHER waves
        const int gw = wave - 4;
        lds_f32* const gred = (lds_f32*)(smem + L.gred);
        uint32_t gbar_n = 0;
        auto gbar = [&](uint32_t code) __attribute__((always_inline)) {
            if (lane == 0) flag_add(flags, F_GBAR);
            ++gbar_n;
            flag_wait(flags, F_GBAR, 4u * gbar_n, code);
        };
        bool dead = false;                                          // this wave gave up on a sweep: later sweeps do not wait either
        // sweep one chunk of 16 x 64 granules (or plain dwords) until every tag matches
        int dbg_layer = 0, dbg_k = -1;                               // timeline slots of the sweep in progress (debug launches)
        auto load_chunk = [&](const u64* src, const uint32_t* plain, int base, int n_gran, unsigned tag, unsigned (&vals)[16], uint32_t code) __attribute__((always_inline)) {
            if (plain != nullptr) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int idx = base + k * 64 + lane;
                    vals[k] = base + k * 64 < n_gran ? ((g_cu32_p)plain)[idx] : 0u;
                }
                return;
            }
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            for (uint32_t it = 0;; ++it) {
                bool okk = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int idx = base + k * 64 + lane;
                    if (base + k * 64 < n_gran) {
                        const u64 x = ld_granule(src + idx);
                        vals[k] = (unsigned)x;
                        okk &= (unsigned)(x >> 32) == tag;
                    } else vals[k] = 0u;
                }
                if (it == 0 && dbg_k >= 0 && gw == 0) stamp(p.dbg, cu, NL, dbg_layer, dbg_k + 1, lane);
                if (__all(okk) || dead) { if (dbg_k >= 0 && gw == 0) { stamp(p.dbg, cu, NL, dbg_layer, dbg_k + 2, lane); if (lane == 0 && p.dbg) p.dbg[((size_t)cu * NL + dbg_layer) * 32 + dbg_k + 5] = it; } return; }
                if (flag_ld(flags, F_ABORT) != 0) { dead = true; return; }
                if ((it & 15u) == 15u && (long long)__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) {
                    ((volatile lds_u32*)flags)[F_ABORT] = code; dead = true; return;
                }
                __builtin_amdgcn_s_sleep(VDD_PL_POLL_SLEEP);
            }
        };
        // gather [M][K] elements (as K/2 column pairs per row) into the swizzled LDS image at `dst`.  lnw != nullptr: RMSNorm on the way -
        // a wave's (<= 2) chunks stay in registers while the four waves exchange their sums of squares, and reach the LDS normalised
        // (rmsnorm_kernel's roundings), so the image is written once.
        uint32_t n_norm = 0;
        auto sweep = [&](const u64* src, const uint32_t* plain, unsigned tag, uint32_t dst, int K, const uint16_t* lnw, uint32_t code) __attribute__((always_inline)) {
            const int Kh = K >> 1, n_gran = M * Kh, n_chunks = (n_gran + 1023) >> 10;
            auto row_of = [&](int gi, int& r, int& cp) __attribute__((always_inline)) {            // gi is wave-uniform: one row per 64 granules
                r = 0;
                if (gi >= Kh) r = 1;
                if (RMAX > 2) { if (gi >= 2 * Kh) r = 2; if (gi >= 3 * Kh) r = 3; }
                cp = gi - r * Kh;
            };
            auto lds_at = [&](int r, int cp) __attribute__((always_inline)) {
                return (lds_u32*)(smem + dst + (uint32_t)r * 2u * (uint32_t)K + (((uint32_t)cp * 4u) ^ ((uint32_t)(r & 15) << 4)));
            };
            if (lnw == nullptr) {
                for (int c = gw; c < n_chunks; c += 4) {
                    unsigned vals[16];
                    load_chunk(src, plain, c * 1024, n_gran, tag, vals, code);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int gi = c * 1024 + k * 64;
                        if (gi < n_gran) { int r, cp; row_of(gi, r, cp); *lds_at(r, cp + lane) = vals[k]; }
                    }
                }
                return;
            }
            const g_cu32_p lnw32 = (g_cu32_p)lnw;
            constexpr bool TWO = RMAX > 2;                          // chunks per wave: 1 (M d <= 8192, the RMAX = 2 instance) or 2 (<= 16384)
            unsigned va[16], vb[TWO ? 16 : 1];
            uint32_t ga[16], gb[TWO ? 16 : 1];
            const int c0 = gw, c1 = TWO ? gw + 4 : n_chunks;
#pragma unroll
            for (int k = 0; k < 16; ++k) {                          // ln weights: in flight under the sweep
                int r, cp; row_of(c0 * 1024 + k * 64, r, cp); ga[k] = c0 * 1024 + k * 64 < n_gran ? lnw32[cp + lane] : 0u;
                if constexpr (TWO) { row_of(c1 * 1024 + k * 64, r, cp); gb[k] = c1 * 1024 + k * 64 < n_gran ? lnw32[cp + lane] : 0u; }
            }
            float rs[RMAX];
#pragma unroll
            for (int r = 0; r < RMAX; ++r) rs[r] = 0.f;
            auto ssq = [&](const unsigned (&v)[16], int c) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int gi = c * 1024 + k * 64;
                    if (gi < n_gran) {
                        int r, cp; row_of(gi, r, cp);
                        const float q = lo(v[k]) * lo(v[k]) + hi(v[k]) * hi(v[k]);
#pragma unroll
                        for (int r2 = 0; r2 < RMAX; ++r2) rs[r2] += r2 == r ? q : 0.f;
                    }
                }
            };
            if (c0 < n_chunks) { load_chunk(src, plain, c0 * 1024, n_gran, tag, va, code); ssq(va, c0); }
            if constexpr (TWO) { if (c1 < n_chunks) { load_chunk(src, plain, c1 * 1024, n_gran, tag, vb, code); ssq(vb, c1); } }
            lds_f32* const gr = gred + (n_norm & 1) * 16;           // two buffers: ONE barrier per normalised sweep
            ++n_norm;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) { const float t = wave_sum(rs[r]); if (lane == 0) gr[gw * 4 + r] = t; }
            gbar(code | 0x10u);
            if (dbg_k >= 0 && gw == 0) stamp(p.dbg, cu, NL, dbg_layer, dbg_k + 3, lane);
            float rstd[RMAX];
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                rstd[r] = rsqrtf(((gr[0 * 4 + r] + gr[1 * 4 + r]) + (gr[2 * 4 + r] + gr[3 * 4 + r])) / (float)K + p.eps);
            auto put = [&](const unsigned (&v)[16], const uint32_t (&gv)[16], int c) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int gi = c * 1024 + k * 64;
                    if (gi < n_gran) {
                        int r, cp; row_of(gi, r, cp);
                        float rsd = rstd[0];
#pragma unroll
                        for (int r2 = 1; r2 < RMAX; ++r2) rsd = r2 == r ? rstd[r2] : rsd;
                        *lds_at(r, cp + lane) = norm_pair(v[k], gv[k], rsd);
                    }
                }
            };
            if (c0 < n_chunks) put(va, ga, c0);
            if constexpr (TWO) { if (c1 < n_chunks) put(vb, gb, c1); }
            if (dbg_k >= 0 && gw == 0) stamp(p.dbg, cu, NL, dbg_layer, dbg_k + 4, lane);
        };

        for (int Lc = 0; Lc < NL; ++Lc) {
            const vdd_layer_desc& ld = p.layers[Lc];
            const unsigned tagL = tag0 + (unsigned)Lc * 8u;
            // ---- G1: the layer's input -> rmsnorm(ln1) -> X
            // (every phase below starts polling global memory only once THIS CU's stream waves are through the producing op: the other CUs
            // are about as far, and four waves sweeping granules through a 27-us gate/up phase cost the weight stream its bandwidth)
            if (Lc > 0) flag_wait(flags, F_SDONE, 4u * (uint32_t)(4 * Lc), 0x610u);
            if (ld.bqkv != nullptr && gw == 3) {                    // qkv bias of this CU's column blocks -> LDS (the stream waves' epilogue
                const int ci = lane;                                // must not load: its loads would queue behind the weight pipeline)
                if (ci < 3 * d / G) ((lds_f32*)(smem + L.bias))[lane] = e2f(((g_u16_p)ld.bqkv)[cu * (3 * d / G) + ci]);
            }
            dbg_layer = Lc; dbg_k = 16;
            if (gw == 0) stamp(p.dbg, cu, NL, Lc, 16, lane);
            sweep(p.g_hpp, Lc == 0 ? reinterpret_cast<const uint32_t*>(p.resid_in) : nullptr, tagL - 8u + T_HPP, L.xd, d, (const uint16_t*)ld.ln1, 0x600u);
            dbg_k = -1;
            if (lane == 0) flag_add(flags, F_XREADY);
            if (gw == 0) stamp(p.dbg, cu, NL, Lc, 11, lane);
            // ---- G2: q, k, v of (row gw, head) -> RoPE -> LDS; the slice that owns the new token writes the KV cache
            flag_wait(flags, F_SDONE, 4u * (uint32_t)(4 * Lc + 1), 0x620u);
            if (cu < H * S && gw < M) {
                const int head = cu / S, sl = cu % S, r = gw;
                const int pp = p.pos[r];
                const float4 c4 = *reinterpret_cast<const float4*>(p.cs + ((size_t)pp * 64 + ((2 * lane) & 63)) * 2);
                const u64* gq = p.g_qkv + (size_t)r * (3 * d / 2) + head * 64 + lane;
                unsigned v3[3];
                const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                for (uint32_t it = 0;; ++it) {
                    bool okk = true;
#pragma unroll
                    for (int w = 0; w < 3; ++w) { const u64 x = ld_granule(gq + (size_t)w * (d / 2)); v3[w] = (unsigned)x; okk &= (unsigned)(x >> 32) == tagL + T_QKV; }
                    if (__all(okk) || dead) break;
                    if (flag_ld(flags, F_ABORT) != 0) { dead = true; break; }
                    if ((it & 15u) == 15u && (long long)__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) {
                        ((volatile lds_u32*)flags)[F_ABORT] = 0x620u; dead = true; break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const float sign = lane < 32 ? -1.f : 1.f;
                auto rope = [&](uint32_t a) __attribute__((always_inline)) {
                    const uint32_t b = __shfl_xor(a, 32);
                    return pack(lo(a) * c4.x + (sign * lo(b)) * c4.y, hi(a) * c4.z + (sign * hi(b)) * c4.w);
                };
                const uint32_t qr = rope(v3[0]), kr = rope(v3[1]);
                lds_u32* qs = (lds_u32*)(smem + L.qkvs + r * 768);
                qs[lane] = qr; qs[64 + lane] = kr; qs[128 + lane] = v3[2];
                if (sl == S - 1) {
                    const size_t o = (size_t)p.slot[r] * p.slot_stride + ((size_t)head * p.t_max + p.cpos[r]) * 128;
                    reinterpret_cast<uint32_t*>((uint16_t*)ld.k_own + o)[lane] = kr;
                    reinterpret_cast<uint32_t*>((uint16_t*)ld.v_own + o)[lane] = v3[2];
                }
            }
            if (lane == 0) flag_add(flags, F_QKVREADY);
            // ---- G3: merge the S key slices of unit (row, head) -> AO granules
            if (cu < H * S) flag_wait(flags, F_ATTARR, 4u * (uint32_t)(Lc + 1), 0x630u);
            for (int j = 0;; ++j) {
                const int u = cu + j * G;
                if (u >= M * H) break;
                if ((j & 3) != gw) continue;
                const int r = u / H, head = u % H;
                const u64* gp = p.g_part + (size_t)(r * H + head) * S * PART_G;
                unsigned a0[8], a1[8], mm_[8], ll_[8];
                const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                for (uint32_t it = 0;; ++it) {
                    bool okk = true;
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        if (s < S) {
                            const u64 x0 = ld_granule(gp + s * PART_G + 2 * lane), x1 = ld_granule(gp + s * PART_G + 2 * lane + 1);
                            const u64 xm = ld_granule(gp + s * PART_G + 128), xl = ld_granule(gp + s * PART_G + 129);
                            a0[s] = (unsigned)x0; a1[s] = (unsigned)x1; mm_[s] = (unsigned)xm; ll_[s] = (unsigned)xl;
                            const unsigned tg = tagL + T_PART;
                            okk &= (unsigned)(x0 >> 32) == tg && (unsigned)(x1 >> 32) == tg && (unsigned)(xm >> 32) == tg && (unsigned)(xl >> 32) == tg;
                        }
                    }
                    if (__all(okk) || dead) break;
                    if (flag_ld(flags, F_ABORT) != 0) { dead = true; break; }
                    if ((it & 15u) == 15u && (long long)__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) {
                        ((volatile lds_u32*)flags)[F_ABORT] = 0x630u; dead = true; break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                float Mg = -INFINITY;
#pragma unroll
                for (int s = 0; s < 8; ++s) if (s < S) Mg = fmaxf(Mg, __builtin_bit_cast(float, mm_[s]));
                float Lg = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s < S) {
                        const float ms = __builtin_bit_cast(float, mm_[s]);
                        const float wgt = ms == -INFINITY ? 0.f : __expf(ms - Mg);          // an empty slice weighs nothing
                        Lg += wgt * __builtin_bit_cast(float, ll_[s]);
                        b0 += wgt * __builtin_bit_cast(float, a0[s]); b1 += wgt * __builtin_bit_cast(float, a1[s]);
                    }
                }
                const float inv = 1.f / Lg;
                st_granule(p.g_ao + (size_t)r * (d / 2) + head * 64 + lane, tagL + T_AO, pack(b0 * inv, b1 * inv));
            }
            if (gw == 0) stamp(p.dbg, cu, NL, Lc, 12, lane);
            // ---- G4: attention output -> X (the qkv projection must have read its last fragment)
            flag_wait(flags, F_SDONE, 4u * (uint32_t)(4 * Lc + 1), 0x640u);
            sweep(p.g_ao, nullptr, tagL + T_AO, L.xd, d, nullptr, 0x640u);
            if (lane == 0) flag_add(flags, F_XREADY);
            if (gw == 0) stamp(p.dbg, cu, NL, Lc, 13, lane);
            // ---- G5: h' -> rmsnorm(ln2) -> X
            flag_wait(flags, F_SDONE, 4u * (uint32_t)(4 * Lc + 2), 0x650u);
            sweep(p.g_hp, nullptr, tagL + T_HP, L.xd, d, (const uint16_t*)ld.ln2, 0x650u);
            if (lane == 0) flag_add(flags, F_XREADY);
            if (gw == 0) stamp(p.dbg, cu, NL, Lc, 14, lane);
            // ---- G6: SwiGLU output -> X_F
            flag_wait(flags, F_SDONE, 4u * (uint32_t)(4 * Lc + 3), 0x660u);
            sweep(p.g_act, nullptr, tagL + T_ACT, L.xf, F, nullptr, 0x660u);
            if (lane == 0) flag_add(flags, F_XREADY);
            if (gw == 0) stamp(p.dbg, cu, NL, Lc, 15, lane);
        }
        if (gw == 0 && lane == 0) {
            // every workgroup read ctrl[0] before the first all-gather, i.e. long before workgroup 0 gets here
            if (cu == 0) __hip_atomic_store((g_u32_p)p.ctrl, launch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // a give-up of any wave of this workgroup -> ctrl[1] (first code wins; the host raises)
    __syncthreads();
    if (tid == 0) {
        const uint32_t ab = ((volatile lds_u32*)flags)[F_ABORT];
        if (ab != 0) atomicCAS(p.ctrl + 1, 0u, ab | ((uint32_t)cu << 16));
    }
}

inline int ok(hipError_t) { return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH; }

constexpr int LDS_CAP = 160 * 1024;

static int n_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}
// geometry of a launch: G workgroups, S key slices per head; 0 rows supported = shape not served
struct Plan { int G, S, max_rows; };
static Plan make_plan(int d, int H, int F, int D, int n_layers) {
    Plan pl{0, 0, 0};
    if (D != 128 || d != H * 128 || d % 256 != 0 || F % 128 != 0 || H < 1) return pl;
    int G = n_cus();
    if (d / 16 < G) G = d / 16;
    if (G < H || G % 2 != 0 || d % G != 0 || F % G != 0 || (d / G) % 2 != 0 || d / G > 32 || 3 * d / G > 64) return pl;
    pl.G = G;
    pl.S = G / H < 8 ? G / H : 8;
    for (int m = 4; m >= 1; --m)
        if (m * d <= 16384 && (int)lds_layout(m, d, F, (m <= 2 && m * d <= 8192) ? 2 : 4, n_layers).total <= LDS_CAP) { pl.max_rows = m; break; }
    return pl;
}
struct WsLayout { size_t qkv, part, ao, hp, act, hpp, total; };
static WsLayout ws_layout(int M, int d, int H, int F, int S) {
    WsLayout w; size_t o = 256;                                   // ctrl words first
    auto take = [&](size_t n_gran) { const size_t at = o; o += (n_gran * 8 + 255) / 256 * 256; return at; };
    w.qkv = take((size_t)M * 3 * d / 2);
    w.part = take((size_t)M * H * S * PART_G);
    w.ao = take((size_t)M * d / 2);
    w.hp = take((size_t)M * d / 2);
    w.act = take((size_t)M * F / 2);
    w.hpp = take((size_t)M * d / 2);
    w.total = o;
    return w;
}

}  // namespace VDD_ELEM_NS
}  // namespace

using namespace VDD_ELEM_NS;

extern "C" {

VDD_HIDDEN int VDD_IMPL(vdd_decode_layers_max_rows)(int d, int H, int F, int D, int n_layers) { return make_plan(d, H, F, D, n_layers).max_rows; }

// partial sums of squares per row in ss_out: one per (workgroup, 16-column block of its d / G columns)
VDD_HIDDEN int VDD_IMPL(vdd_decode_layers_ss_cols)(int d, int H, int F, int D) {
    const Plan pl = make_plan(d, H, F, D, 1);
    return pl.max_rows < 1 ? 0 : pl.G * ((d / pl.G + 15) / 16);
}

VDD_HIDDEN int64_t VDD_IMPL(vdd_decode_layers_workspace_bytes)(int M, int d, int H, int F, int D) {
    const Plan pl = make_plan(d, H, F, D, 1);
    if (pl.max_rows < 1 || M < 1 || M > pl.max_rows) return 0;
    return (int64_t)ws_layout(M, d, H, F, pl.S).total;
}

VDD_HIDDEN int VDD_IMPL(vdd_decode_layers)(const vdd_layer_desc* layers, int n_layers, const void* resid_in, void* resid_out, float* ss_out,
                                          const int32_t* pos, const int32_t* cpos, const int32_t* slot, const float* cos_sin, const int32_t* rows,
                                          int M, int d, int H, int Hkv, int F, int D, float eps, float scale, int64_t slot_stride, int t_max,
                                          int64_t prefix_stride, int prefix_tmax, int has_qkv_bias, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M <= 0 || n_layers <= 0) return VDD_OK;
    if (!layers || !resid_in || !resid_out || !ss_out || !pos || !cpos || !slot || !cos_sin || !rows || !workspace) return VDD_ERR_INVALID_ARG;
    if (Hkv != H || n_layers > 120) return VDD_ERR_UNSUPPORTED;
    const Plan pl = make_plan(d, H, F, D, n_layers);
    if (pl.max_rows < 1 || M > pl.max_rows) return VDD_ERR_UNSUPPORTED;
    const WsLayout w = ws_layout(M, d, H, F, pl.S);
    if (workspace_bytes < (int64_t)w.total || ((uintptr_t)workspace & 255) != 0) return VDD_ERR_INVALID_ARG;
    unsigned char* ws = (unsigned char*)workspace;
    Params p;
    p.layers = layers; p.n_layers = n_layers; p.M = M; p.d = d; p.H = H; p.F = F; p.S = pl.S; p.G = pl.G;
    p.eps = eps; p.scale = scale; p.has_bias = has_qkv_bias ? 1 : 0;
    p.resid_in = (const uint16_t*)resid_in; p.resid_out = (uint16_t*)resid_out; p.ss_out = ss_out;
    p.pos = pos; p.cpos = cpos; p.slot = slot; p.cs = cos_sin; p.rows = (const AttnRow*)rows;
    p.slot_stride = slot_stride; p.pre_stride = prefix_stride; p.t_max = t_max; p.pre_tmax = prefix_tmax;
    p.ctrl = (unsigned*)ws;
    p.g_qkv = (u64*)(ws + w.qkv); p.g_part = (u64*)(ws + w.part); p.g_ao = (u64*)(ws + w.ao); p.g_hp = (u64*)(ws + w.hp);
    p.g_act = (u64*)(ws + w.act); p.g_hpp = (u64*)(ws + w.hpp);
    // a workspace with room for it behind the exchange buffers gets the [G][n_layers][16] timeline (tools/persistent_probe.py)
    p.dbg = workspace_bytes >= (int64_t)(w.total + (size_t)pl.G * n_layers * 256) ? (u64*)(ws + w.total) : nullptr;
    const int rmax = (M <= 2 && M * d <= 8192) ? 2 : 4;          // the 2-row instance keeps ONE 1024-granule chunk per gather wave in registers
    const size_t lds = lds_layout(M, d, F, rmax, n_layers).total;
#define VDD_LAYERS(R)                                                                                                              \
    do {                                                                                                                           \
        static bool attr = false;                                                                                                  \
        if (!attr) { (void)hipFuncSetAttribute((const void*)decode_layers_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CAP); attr = true; } \
        hipLaunchKernelGGL((decode_layers_kernel<R>), dim3(pl.G), dim3(512), lds, (hipStream_t)stream, p);                         \
    } while (0)
    if (rmax == 2) VDD_LAYERS(2); else VDD_LAYERS(4);
#undef VDD_LAYERS
    return ok(hipSuccess);
}

}  // extern "C"
