// Fused visual-contrastive-decoding sampling tail for gfx950 (MI355X, wave64).
//
// One launch replaces the ~15 eager kernels + 2 host syncs of the reference per decode
// step (vcd_utils/vcd_sample.py:185-207,257-260,285-288):
//
//   c   = fl(fl(c+d)/2)                         both-branch average            (:185)
//   cut = fl(max_j v_j + log beta)              adaptive-plausibility cutoff   (:191)
//   x   = fl(fl((1+a) v) - fl(a c)) ; x[v<cut] = -inf                          (:193-194)
//   x   = fl(x / T) ; top-k ; top-p             HF warpers                     (:198)
//   tok ~ softmax(x)                            multinomial                    (:201-202)
//   tok = tok*unfinished + pad*(1-unfinished) ; unfinished *= (tok not in eos) (:260,:286)
//
// fl() = round-to-nearest-even into the MODEL dtype after every torch op, exactly as
// eager torch does, so the scores row is bit-identical to the reference's.
//
// Mapping: one 512-thread workgroup (8 waves) per row; the row is cut into 16-byte
// chunks and chunk ch belongs to thread ch % 512 in EVERY pass, so a wave reads 1 KiB
// contiguous per load and no pass ever reads another thread's chunk (no data barriers,
// only the 8-entry reduction exchanges).  The working row lives on chip: the first chunks
// of every thread in LDS, its last three in registers (V=32000 bf16: 40 KiB of LDS + 12
// VGPRs -> THREE workgroups per CU; rows in flight per CU, not threads per row, is what
// the kernel's throughput follows); v is read from HBM once, c/d once (only where a
// candidate survives), scores written once.  Rows too large for that (Qwen, V=151936)
// keep the working row in the caller's scores/workspace buffer instead (re-reads served
// by L2 / Infinity Cache).
// Every later pass (softmax statistics, radix threshold selection for top-k / top-p,
// inverse-CDF sampling, top-n extraction) runs over the LDS row.
//
// Roofline: HBM-bound; nothing here is a contraction, so no MFMA.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <string.h>
#include <stdio.h>

#include "vdd_hip.h"

namespace {

constexpr int BLOCK = 512;               // threads per row: 3 rows in flight per CU instead of 2
constexpr int NWAVE = BLOCK / 64;
constexpr int NCOPY = BLOCK / 256;       // copies of the selection histograms (one per 256 threads)
constexpr int NREG = 3;                  // chunks per thread held in REGISTERS instead of the LDS row
constexpr int UNR = 4;                   // chunks in flight per thread per batch
constexpr int LDS_ROW_BYTES_MAX = 140 * 1024;       // + Smem + the top-p candidate list stay inside the CU's 160 KiB

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_s;

// ------------------------------------------------------------------ dtype traits
template <int DT> struct Tr;
template <> struct Tr<VDD_F16> {
    using bits_t = uint16_t;
    static constexpr int KEYBITS = 16, EPC = 8;
    static __device__ __forceinline__ float to_f(uint32_t b) { return (float)__builtin_bit_cast(_Float16, (uint16_t)b); }
    static __device__ __forceinline__ uint32_t from_f(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    static constexpr uint32_t NEG_INF = 0xFC00u;
};
template <> struct Tr<VDD_BF16> {
    using bits_t = uint16_t;
    static constexpr int KEYBITS = 16, EPC = 8;
    static __device__ __forceinline__ float to_f(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
    // v_cvt_pk_bf16_f32: round to nearest even in one instruction (the shift / add / select form costs 7 per element, and the
    // contrast arithmetic rounds four times per element like the reference's bf16 tensors do)
    static __device__ __forceinline__ uint32_t from_f(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
    static constexpr uint32_t NEG_INF = 0xFF80u;
};
template <> struct Tr<VDD_F32> {
    using bits_t = uint32_t;
    static constexpr int KEYBITS = 32, EPC = 4;
    static __device__ __forceinline__ float to_f(uint32_t b) { return __builtin_bit_cast(float, b); }
    static __device__ __forceinline__ uint32_t from_f(float f) { return __builtin_bit_cast(uint32_t, f); }
    static constexpr uint32_t NEG_INF = 0xFF800000u;
};

template <int DT> __device__ __forceinline__ float rnd(float f) { return Tr<DT>::to_f(Tr<DT>::from_f(f)); }

// order-preserving key: larger float <=> larger unsigned key
template <int DT> __device__ __forceinline__ uint32_t okey(uint32_t b) {
    if constexpr (Tr<DT>::KEYBITS == 16) return (b & 0x8000u) ? (~b & 0xFFFFu) : (b | 0x8000u);
    else return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <int DT> __device__ __forceinline__ uint32_t getb(const uint32_t* w, int j) {
    if constexpr (Tr<DT>::KEYBITS == 16) return (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
    else return w[j];
}
template <int DT> __device__ __forceinline__ void setb(uint32_t* w, int j, uint32_t b) {
    if constexpr (Tr<DT>::KEYBITS == 16) {
        if (j & 1) w[j >> 1] = (w[j >> 1] & 0x0000FFFFu) | (b << 16);
        else w[j >> 1] = (w[j >> 1] & 0xFFFF0000u) | (b & 0xFFFFu);
    } else w[j] = b;
}

// ------------------------------------------------------------------ kernel params (by value)
struct KP {
    int kl;                    // chunks per thread kept in LDS; the remaining (<= NREG) ones live in registers
    const void* v; const void* c; const void* d;
    long long sv, sc, sd, ss, sw, st;
    int B, V;
    unsigned flags;
    int min_keep, top_k, n_eos, n_top;
    float s1, s2, log_beta, temp, inv_temp, one_minus_p;   // s1 = 1+alpha, s2 = alpha (fp32 scalars, as torch passes them)
    int use_temp, use_topp;
    unsigned long long seed, offset;
    const float* uniforms;
    const unsigned long long* offset_ptr;
    const long long* eos;
    long long pad;
    long long* unfinished;
    long long* next_tokens;
    void* scores;      // optional output
    void* work;        // global working row (== scores when given) — only used when the row does not fit LDS
    float* top_prob; long long* top_tok;
    int* status;
    int vec_in, vec_out, vec_work;   // 16-B aligned fast paths usable
    // logits-processor stage between contrast and warp (vcd_sample.py:197)
    const int* eos_min;              // [B] or null: eos ids -> -inf while step < eos_min[row]
    long long step; const long long* step_ptr;
    const int* force;                // [B] or null: scores[row, force_id] = force_val where force[row] != 0
    long long force_id; float force_val;
};

// ------------------------------------------------------------------ block collectives
struct Smem {
    float f[2][NWAVE];
    int i[2][NWAVE];
    union {                    // NCOPY copies (wave % NCOPY) to thin same-address atomic contention; the count-based and
        unsigned hist[NCOPY][256];      // the mass-based selection never run at the same time
        float histf[NCOPY][256];
    };
    unsigned sel[4];
    float self[2];
    // compact candidate list of the single-wave tail (rows with <= 64 finite scores)
    unsigned cand_n;
    unsigned live_n;           // live chunks of pass B appended to the LiveList so far
    unsigned cand_rank[64];
    float cand_x[64];
    int cand_idx[64];
};

// Exact HF top-p (model-dtype softmax -> sequential cumulative sum in ascending (value, index) order -> cum <= fl(1-p)) needs the
// candidates as an explicit list: up to TOPP_EXACT_MAX of them (all golden vectors; a row that keeps more after top-k falls
// back to integrating the fp32 mass by radix selection).  Carved behind Smem only when a launch has top-p on.
constexpr int TOPP_EXACT_MAX = 1024;
struct ToppScratch { float gx[TOPP_EXACT_MAX]; int gi[TOPP_EXACT_MAX]; float sp[TOPP_EXACT_MAX]; };

// Pass B's compacted work list (same place as ToppScratch, which is only used after it): chunk index + the v chunk in, the
// contrasted chunk out.  448 entries keep three 32000-wide bf16 rows per CU (40 KiB LDS row part + Smem + the list each).
constexpr int LIVE_CAP = 448;
struct LiveList { uint4 data[LIVE_CAP]; int ch[LIVE_CAP]; };
constexpr int FAST_TAIL_MAX = 64;       // candidates the single-wave tail holds (one per lane); 0 sends every row through the block-wide tail
                                         // (how round 4 priced that tail: +7 us per launch on one-survivor rows, DESIGN_APPENDIX.md)
constexpr int TOPK_LIST_MAX = 1024;      // ordered keys of the top-k candidate list (same scratch region)
static_assert(TOPK_LIST_MAX * 4 <= (int)sizeof(LiveList), "");

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sumi(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// max and two integer sums in one exchange
__device__ __forceinline__ void block_max_count2(float& m, int& c0, int& c1, Smem& sm, int lane, int wave) {
    m = wave_max(m); c0 = wave_sumi(c0); c1 = wave_sumi(c1);
    if (lane == 0) { sm.f[0][wave] = m; sm.i[0][wave] = c0; sm.i[1][wave] = c1; }
    __syncthreads();
    float mm = sm.f[0][0]; int a = sm.i[0][0], b = sm.i[1][0];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) { mm = fmaxf(mm, sm.f[0][w]); a += sm.i[0][w]; b += sm.i[1][w]; }
    __syncthreads();
    m = mm; c0 = a; c1 = b;
}
__device__ __forceinline__ float block_sum(float v, Smem& sm, int lane, int wave) {
    v = wave_sum(v);
    if (lane == 0) sm.f[1][wave] = v;
    __syncthreads();
    float s = sm.f[1][0];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) s += sm.f[1][w];
    __syncthreads();
    return s;
}

// philox4x32-10, returns a 24-bit uniform in [0,1)
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned long long offset, unsigned row) {
    uint32_t c0 = (uint32_t)offset, c1 = (uint32_t)(offset >> 32), c2 = row, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------ chunk loads / stores (global)
// A chunk is 16 bytes = EPC elements held as 4 packed words.
// STREAM: the chunk is read once by the whole launch (logits in, the LDS-resident row): nontemporal, it takes no L2 line from anyone
template <int DT, bool STREAM = false>
__device__ __forceinline__ void gload(const void* base, long long row_off, int ch, int V, int vec, uint32_t* w) {
    using B = typename Tr<DT>::bits_t;
    constexpr int EPC = Tr<DT>::EPC;
    const B* p = reinterpret_cast<const B*>(base) + row_off;
    const int idx0 = ch * EPC;
    if (vec && idx0 + EPC <= V) {
        uint4 a;
        if constexpr (STREAM) a = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_s*>(p + idx0)));
        else a = *reinterpret_cast<const uint4*>(p + idx0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            uint32_t b = (idx0 + j < V) ? (uint32_t)p[idx0 + j] : Tr<DT>::NEG_INF;
            setb<DT>(w, j, b);
        }
    }
}
template <int DT, bool STREAM = false>
__device__ __forceinline__ void gstore(void* base, long long row_off, int ch, int V, int vec, const uint32_t* w) {
    using B = typename Tr<DT>::bits_t;
    constexpr int EPC = Tr<DT>::EPC;
    B* p = reinterpret_cast<B*>(base) + row_off;
    const int idx0 = ch * EPC;
    if (vec && idx0 + EPC <= V) {
        if constexpr (STREAM) __builtin_nontemporal_store(u32x4_s{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4_s*>(p + idx0));
        else *reinterpret_cast<uint4*>(p + idx0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for (int j = 0; j < EPC; ++j) if (idx0 + j < V) p[idx0 + j] = (B)getb<DT>(w, j);
    }
}

// The working row.  LDSROW: chunk k of a thread (ch = tid + k * BLOCK) lives in LDS for k < kl and in one of the thread's
// NREG register chunks for k >= kl - a 64 KB row would allow two workgroups per CU; with the last three chunks of every
// thread in registers (12 VGPRs) the LDS part of a V = 32000 bf16 row is 40 KB and three fit.  !LDSROW: the caller's
// global scores / workspace buffer (rows too large for LDS).
template <int DT, bool LDSROW>
struct Row {
    uint4* lds; void* g; long long goff; int V; int vec; int kl; uint32_t (*rc)[4];
    __device__ __forceinline__ void get(int ch, uint32_t* w) const {
        if constexpr (LDSROW) {
            const int r = ch / BLOCK - kl;
            if (r < 0) { uint4 a = lds[ch]; w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; }
            else if (r == 0) { w[0] = rc[0][0]; w[1] = rc[0][1]; w[2] = rc[0][2]; w[3] = rc[0][3]; }
            else if (r == 1) { w[0] = rc[1][0]; w[1] = rc[1][1]; w[2] = rc[1][2]; w[3] = rc[1][3]; }
            else { w[0] = rc[2][0]; w[1] = rc[2][1]; w[2] = rc[2][2]; w[3] = rc[2][3]; }
        } else gload<DT>(g, goff, ch, V, vec, w);
    }
    __device__ __forceinline__ void put(int ch, const uint32_t* w) const {
        if constexpr (LDSROW) {
            const int r = ch / BLOCK - kl;
            if (r < 0) lds[ch] = make_uint4(w[0], w[1], w[2], w[3]);
            else if (r == 0) { rc[0][0] = w[0]; rc[0][1] = w[1]; rc[0][2] = w[2]; rc[0][3] = w[3]; }
            else if (r == 1) { rc[1][0] = w[0]; rc[1][1] = w[1]; rc[1][2] = w[2]; rc[1][3] = w[3]; }
            else { rc[2][0] = w[0]; rc[2][1] = w[1]; rc[2][2] = w[2]; rc[2][3] = w[3]; }
        } else gstore<DT>(g, goff, ch, V, vec, w);
    }
};

// ------------------------------------------------------------------ radix threshold selection
// Count-based: ordered key of the k-th largest among finite entries (k <= #finite).
template <int DT, bool L>
__device__ __forceinline__ uint32_t select_kth_key(const Row<DT, L>& R, int nch, unsigned k, Smem& sm, int tid, int lane, int wave,
                                                   unsigned long long live) {
    constexpr int KB = Tr<DT>::KEYBITS, EPC = Tr<DT>::EPC;
    uint32_t prefix = 0, pmask = 0;
    unsigned rem = k;
    for (int shift = KB - 8; shift >= 0; shift -= 8) {
        sm.hist[tid >> 8][tid & 255] = 0;        // BLOCK / 256 = NCOPY copies
        __syncthreads();
        unsigned* h = sm.hist[wave % NCOPY];
        for (int ch = tid, kq = 0; ch < nch; ch += BLOCK, ++kq) {
            if (kq < 64 && ((live >> kq) & 1ull) == 0ull) continue;        // chunk without a finite score (never written in sparse mode)
            uint32_t w[4]; R.get(ch, w);
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                uint32_t b = getb<DT>(w, j);
                if (b != Tr<DT>::NEG_INF) {
                    uint32_t key = okey<DT>(b);
                    if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & 0xFFu], 1u);
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            unsigned c[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int bin = 4 * lane + q; c[q] = 0;
#pragma unroll
                for (int cp = 0; cp < NCOPY; ++cp) c[q] += sm.hist[cp][bin];
            }
            unsigned loc = c[0] + c[1] + c[2] + c[3];
            unsigned suf = loc;   // inclusive suffix sum over lanes >= lane
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_down(suf, o); if (lane + o < 64) suf += t; }
            unsigned above = suf - loc;
            if (above < rem && rem <= suf) {
                unsigned a = above; bool done = false;
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    if (!done) {
                        if (a + c[q] >= rem) { sm.sel[0] = 4 * lane + q; sm.sel[1] = rem - a; done = true; }
                        else a += c[q];
                    }
                }
            }
        }
        __syncthreads();
        prefix |= sm.sel[0] << shift;
        pmask |= 0xFFu << shift;
        rem = sm.sel[1];
        __syncthreads();
    }
    return prefix;
}

// Mass-based, ascending: smallest key K such that mass(key <= K) > thr.  Entries with
// key < K are the ones HF's TopPLogitsWarper removes (cum <= 1-p).  Returns false if the
// whole row's mass never exceeds thr (then only min_keep survive).
template <int DT, bool L>
__device__ __forceinline__ bool select_mass_key(const Row<DT, L>& R, int nch, float m, float thr, Smem& sm,
                                                int tid, int lane, int wave, uint32_t& out_key, unsigned long long live,
                                                int digits_known = 0, uint32_t prefix = 0, float below = 0.f) {
    // digits_known = 1: the caller found the top digit (`prefix`) and the mass under it (`below`) without the histogram
    constexpr int KB = Tr<DT>::KEYBITS, EPC = Tr<DT>::EPC;
    uint32_t pmask = digits_known ? 0xFFu << (KB - 8) : 0u;
    bool ok = true;
    for (int shift = KB - 8 - 8 * digits_known; shift >= 0; shift -= 8) {
        sm.histf[tid >> 8][tid & 255] = 0.f;
        __syncthreads();
        float* h = sm.histf[wave % NCOPY];
        for (int ch = tid, kq = 0; ch < nch; ch += BLOCK, ++kq) {
            if (kq < 64 && ((live >> kq) & 1ull) == 0ull) continue;        // chunk without a finite score (never written in sparse mode)
            uint32_t w[4]; R.get(ch, w);
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                uint32_t b = getb<DT>(w, j);
                if (b != Tr<DT>::NEG_INF) {
                    uint32_t key = okey<DT>(b);
                    if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & 0xFFu], __expf(Tr<DT>::to_f(b) - m));
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            float c[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int bin = 4 * lane + q; c[q] = 0.f;
#pragma unroll
                for (int cp = 0; cp < NCOPY; ++cp) c[q] += sm.histf[cp][bin];
            }
            float loc = (c[0] + c[1]) + (c[2] + c[3]);
            float pre = loc;   // inclusive prefix over lanes <= lane
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(pre, o); if (lane >= o) pre += t; }
            unsigned long long cross = __ballot(below + pre > thr);
            unsigned long long nonempty = __ballot(loc > 0.f);
            int hit_lane;
            if (cross) hit_lane = __ffsll((long long)cross) - 1;
            else if (shift == KB - 8) hit_lane = -1;                               // whole row below thr (top digit only)
            else hit_lane = nonempty ? 63 - __clzll((long long)nonempty) : -1;     // association fuzz: top non-empty
            if (hit_lane < 0) { if (lane == 0) sm.sel[2] = 0; }
            else if (lane == hit_lane) {
                float a = below + (pre - loc);
                int bin = -1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (bin < 0) { if (a + c[q] > thr) bin = q; else a += c[q]; }
                }
                if (bin < 0) {            // fuzz: take the highest non-empty bin of this lane
                    a = below + (pre - loc);
                    bin = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (c[q] > 0.f) bin = q;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (q < bin) a += c[q];
                }
                sm.sel[0] = 4 * lane + bin; sm.self[0] = a; sm.sel[2] = 1;
            }
        }
        __syncthreads();
        ok = sm.sel[2] != 0;
        uint32_t bin = sm.sel[0]; float nb = sm.self[0];
        __syncthreads();
        if (!ok) break;
        prefix |= bin << shift;
        pmask |= 0xFFu << shift;
        below = nb;
    }
    out_key = prefix;
    return ok;
}

template <int DT, bool L>
__device__ __forceinline__ void mask_below_key(const Row<DT, L>& R, int nch, uint32_t thr_key, int tid, unsigned long long live, int thr_idx = 0) {
    // removes every element below (thr_key, thr_idx) in (value, index) order: thr_idx = 0 is the plain value threshold
    constexpr int EPC = Tr<DT>::EPC;
    for (int ch = tid, kq = 0; ch < nch; ch += BLOCK, ++kq) {
        if (kq < 64 && ((live >> kq) & 1ull) == 0ull) continue;        // chunk without a finite score (never written in sparse mode)
        uint32_t w[4]; R.get(ch, w);
        bool changed = false;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            uint32_t b = getb<DT>(w, j);
            if (b == Tr<DT>::NEG_INF) continue;
            const uint32_t k = okey<DT>(b);
            if (k < thr_key || (k == thr_key && ch * EPC + j < thr_idx)) { setb<DT>(w, j, Tr<DT>::NEG_INF); changed = true; }
        }
        if (changed) R.put(ch, w);
    }
}

// ------------------------------------------------------------------ the kernel
template <int DT, bool LDSROW, bool PROC>
// Waves per SIMD: 6 = three workgroups per CU for the rows that live in LDS (40 KiB each at V = 32000: LDS is the limit anyway).  Rows that
// stay in global memory (V = 151,936) hold no LDS row, and three workgroups per CU made 768 rows a round: 1,024 rows cost two.  Their
// plain instance is built for 8 waves per SIMD (64 registers, 18 - 70 spilled into the L1-resident scratch): four workgroups per CU,
// V = 151,936 without scores at 1,024 rows 105 -> 94 us, at 3,072 rows 250 -> 237 us (tools/kernel_points_qwen.py).  The processor
// instance of those rows would spill 270 registers: it keeps 6.
__global__ void __launch_bounds__(BLOCK, (LDSROW || PROC) ? 6 : 8) vdd_contrast_sample_kernel(KP p) {
    constexpr int EPC = Tr<DT>::EPC;
    constexpr uint32_t NINF = Tr<DT>::NEG_INF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int V = p.V;
    const int nch = (V + EPC - 1) / EPC;
    const int lds_chunks = LDSROW ? (nch < p.kl * BLOCK ? nch : p.kl * BLOCK) : 0;
    Smem& sm = *reinterpret_cast<Smem*>(smem_raw + (size_t)lds_chunks * 16);
    uint32_t rc[NREG][4];
    Row<DT, LDSROW> R{reinterpret_cast<uint4*>(smem_raw), p.work, (long long)row * p.sw, V, p.vec_work, p.kl, rc};
    const long long ov = (long long)row * p.sv, oc = (long long)row * p.sc, od = (long long)row * p.sd;

    float m = -INFINITY; int nfin = 0; int t_nan = 0, t_pinf = 0;
    const bool recip = (p.flags & VDD_TEMP_RECIPROCAL) != 0;
    if (tid == 0) { sm.cand_n = 0u; sm.live_n = 0u; }
    // Bit k set <=> this thread's k-th chunk (ch = tid + k * BLOCK) may hold a finite score.  After the plausibility mask a
    // row keeps a handful of candidates, and every pass over the working row costs ~10 VALU instructions per ELEMENT
    // (measured: 2.7 us per row for the mass pass alone, the passes after the scores store were 1/3 of the kernel), so
    // the passes of the sampling tail walk only the flagged chunks.  Chunks past bit 63 (V > 2^19) count as flagged.
    unsigned long long livemask = 0ull;
    // Logits-processor stage (vcd_sample.py:197, between contrast and warp), block-uniform per row:
    //   bit 0  every eos id -> -inf  (HF MinLength / MinNewTokensLength processors: the row has not produced enough tokens yet)
    //   bit 1  scores[force_id] = force_val (Qwen StopWordsLogitsProcessor, qwen_generation_utils.py:352-359: a stop sequence
    //          matched; applied AFTER bit 0 as in HF's processor order: defaults first, custom processors last)
    // PROC is a template parameter: launches without a processor stage (the roofline shape among them) run the instance that
    // has none of this code - as a run-time flag the stage cost the plain contrast launch 7 % (118 vs 110 us at B = 4096).
    int proc = 0;
    if constexpr (PROC) {
        if (p.eos_min != nullptr && p.n_eos > 0) {
            const long long s_now = p.step + (p.step_ptr ? *p.step_ptr : 0ll);
            proc |= (s_now < (long long)p.eos_min[row]) ? 1 : 0;
        }
        if (p.force != nullptr && p.force[row] != 0) proc |= 2;
    }
    const int force_ch = (PROC && (proc & 2)) ? (int)(p.force_id / EPC) : -1;
    auto is_eos = [&](int idx) { bool e = false; for (int q = 0; q < p.n_eos; ++q) e |= (long long)idx == p.eos[q]; return e; };

    if (p.c != nullptr) {
        // ---- pass A: v -> working row, row max (vcd_sample.py:191) --------------------
        float vmax = -INFINITY; int vnan = 0, zero = 0;
        LiveList& LL = *reinterpret_cast<LiveList*>(reinterpret_cast<unsigned char*>(&sm) + ((sizeof(Smem) + 15) & ~(size_t)15));
        // Rows too large for the chip (!LDSROW: V = 151,936) used to read v TWICE - once for the row maximum, once more (from L2 / the
        // Infinity Cache) to contrast and mask.  Now pass A reads v ONCE, nontemporally, and stashes every chunk that CAN hold a
        // survivor in the LDS work list of pass B: a chunk is kept when one of its elements reaches fl(bm + log beta), bm = the
        // block-wide maximum of all batches read so far (of this very batch for the first one).  bm <= the final maximum and the
        // rounding is monotone, so that cutoff is <= the final one: the list is a superset of the live chunks, and pass B contrasts
        // list entries only (chunks that fail the final cutoff come out all -inf).  A list overflow (a flat row, or a maximum that
        // shows up late in a row of large values) falls back to the two-read form.
        [[maybe_unused]] unsigned long long stash = 0ull;     // this thread's stashed chunks (bit k: ch = tid + k * BLOCK)
        [[maybe_unused]] bool stash_ok = !LDSROW && nch <= 64 * BLOCK;
        [[maybe_unused]] const float lb_a = (p.flags & VDD_CUTOFF_F32_SCALAR) ? p.log_beta : rnd<DT>(p.log_beta);
        float bm = -INFINITY;
        int kb_a = 0;
        if constexpr (!LDSROW) {        // the published wave maxima of the stash cut (the histogram's place: not in use before the selection passes)
            if (tid < NWAVE) sm.histf[0][tid] = -INFINITY;
            __syncthreads();
        }
        for (int base = tid; LDSROW ? base < nch : base - tid < nch; base += UNR * BLOCK, kb_a += UNR) {   // (!LDSROW: uniform trip count, block-wide exchanges inside)
            uint32_t q[UNR][4];
#pragma unroll
            for (int u = 0; u < UNR; ++u) { const int ch = base + u * BLOCK; if (ch < nch) gload<DT, true>(p.v, ov, ch, V, p.vec_in, q[u]); }
            [[maybe_unused]] float cmax[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ch = base + u * BLOCK;
                if constexpr (!LDSROW) cmax[u] = -INFINITY;
                if (ch < nch) {
                    if constexpr (LDSROW) R.put(ch, q[u]);
#pragma unroll
                    for (int j = 0; j < EPC; ++j) {
                        float f = Tr<DT>::to_f(getb<DT>(q[u], j)); vnan |= (f != f) ? 1 : 0; vmax = fmaxf(vmax, f);
                        if constexpr (!LDSROW) { if (ch * EPC + j < V) cmax[u] = fmaxf(cmax[u], f); }
                    }
                }
            }
            if constexpr (!LDSROW) {
                if (stash_ok) {
                    // a LOWER BOUND of the block-wide maximum so far: this wave's own running maximum and whatever the other waves have
                    // published by now - no barrier (any value ever stored in lag[] is a maximum over elements of this row, hence <= the
                    // final one, which is all the superset argument above needs; WHICH chunks get stashed beyond the live ones may then differ
                    // from run to run, the result cannot).  With two barriers per batch the loads of a batch only went out after the
                    // previous one had been exchanged: one memory round trip per 16K elements, 3.0 TB/s at V = 151,936.
                    const float t = wave_max(vmax);
                    volatile float* lag = sm.histf[0];
                    if (lane == 0) lag[wave] = t;
                    float nb = t;
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) nb = fmaxf(nb, lag[w]);
                    bm = fmaxf(bm, nb);
                    const float cut_lag = rnd<DT>(__fadd_rn(bm, lb_a));
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int ch = base + u * BLOCK;
                        if (ch < nch && (!(cmax[u] < cut_lag) || (PROC && ch == force_ch))) {
                            const unsigned slot = atomicAdd(&sm.live_n, 1u);
                            if (slot < (unsigned)LIVE_CAP) {
                                LL.ch[slot] = ch;
                                LL.data[slot] = make_uint4(q[u][0], q[u][1], q[u][2], q[u][3]);
                                stash |= 1ull << (kb_a + u);
                            }
                        }
                    }
                }
            }
        }
        block_max_count2(vmax, vnan, zero, sm, lane, wave);
        if constexpr (!LDSROW) {
            if (stash_ok && sm.live_n > (unsigned)LIVE_CAP) stash_ok = false;       // (uniform: read behind the barriers of the exchange above)
            __syncthreads();
            if (!stash_ok && tid == 0) sm.live_n = 0u;
            __syncthreads();
        }
        t_nan = vnan ? 1 : 0;
        const float lb = (p.flags & VDD_CUTOFF_F32_SCALAR) ? p.log_beta : rnd<DT>(p.log_beta);
        const float cutoff = rnd<DT>(__fadd_rn(vmax, lb));                                   // :191
        const bool both = p.d != nullptr;
        // ---- pass B: contrast + mask + temperature -> working row ----------------------
        // Where v < cutoff the result is -inf whatever c/d hold (:194), so the contrast
        // branches are only READ for chunks that contain a survivor: with the usual beta the
        // c/d rows cost a few 64-B sectors instead of V*e bytes each.
        // A chunk with a survivor costs ~300 VALU instructions (8 elements x four roundings, an IEEE division, the mask),
        // and run where the chunk is found it runs for ONE live lane of a wave at the price of 64: at 12 survivors per row that
        // was +30 % on the launch, at 80 +105 %.  So the live chunks of the row are COMPACTED first: every thread appends its
        // live chunks (index + the v chunk) to an LDS list, then the list is processed one entry per thread with all lanes busy
        // (the c / d loads of a pass go out together), and the results are copied back into the row: by any thread for the LDS
        // (or global) part, by the owner for its register-resident chunks.  Past LIVE_CAP entries a chunk is done in place.
        auto contrast_chunk = [&](auto both_c, int ch, const uint32_t* qv, const uint32_t* qc, const uint32_t* qd, uint32_t* x4) {
            constexpr bool BOTH = decltype(both_c)::value;
            if constexpr (Tr<DT>::KEYBITS == 16) { x4[0] = x4[1] = x4[2] = x4[3] = NINF | (NINF << 16); }
            else { x4[0] = x4[1] = x4[2] = x4[3] = NINF; }
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const int idx = ch * EPC + j;
                float vf = Tr<DT>::to_f(getb<DT>(qv, j)), cf = Tr<DT>::to_f(getb<DT>(qc, j));
                if constexpr (BOTH) {
                    float df = Tr<DT>::to_f(getb<DT>(qd, j));
                    cf = rnd<DT>(__fmul_rn(rnd<DT>(__fadd_rn(cf, df)), 0.5f));                 // :185
                }
                float a = rnd<DT>(__fmul_rn(vf, p.s1));
                float b = rnd<DT>(__fmul_rn(cf, p.s2));
                float x = rnd<DT>(__fsub_rn(a, b));                                           // :193
                bool masked = (vf < cutoff) || (idx >= V);                                    // :194
                if (PROC && proc) {                                                           // :197 logits_processor
                    if ((proc & 1) && is_eos(idx)) masked = true;
                    if ((proc & 2) && (long long)idx == p.force_id) { x = rnd<DT>(p.force_val); masked = false; }
                }
                if (p.use_temp) x = rnd<DT>(recip ? __fmul_rn(x, p.inv_temp) : __fdiv_rn(x, p.temp));   // HF temperature
                if (!masked) {
                    setb<DT>(x4, j, Tr<DT>::from_f(x));
                    nfin += 1; m = fmaxf(m, x); t_nan |= (x != x) ? 1 : 0; t_pinf |= (x == INFINITY) ? 1 : 0;
                }
            }
        };
        auto pass_b = [&](auto both_c) {
            constexpr bool BOTH = decltype(both_c)::value;
            uint32_t ninf4[4];
            if constexpr (Tr<DT>::KEYBITS == 16) { ninf4[0] = ninf4[1] = ninf4[2] = ninf4[3] = NINF | (NINF << 16); }
            else { ninf4[0] = ninf4[1] = ninf4[2] = ninf4[3] = NINF; }
            // a global working row that nobody asked for (no scores row) only ever gets its LIVE chunks: every later pass skips
            // the chunks without a livemask bit, so the -inf chunks need not exist (V = 151,936: the [B, V] round trip was half the launch)
            const bool sparse = !LDSROW && p.scores == nullptr && nch <= 64 * BLOCK;
            int rslot[NREG];                       // list slots of this thread's register-resident live chunks
#pragma unroll
            for (int r = 0; r < NREG; ++r) rslot[r] = -1;
            int kb = 0;
            bool walk = true;
            if constexpr (!LDSROW) {
                if (stash_ok) {                // single-read form: the list already holds every chunk that can be live
                    walk = false;
                    livemask = stash;
                    if (!sparse) {             // the scores / working row: -inf everywhere the list does not cover (written by the owner)
                        int k2 = 0;
                        for (int ch = tid; ch < nch; ch += BLOCK, ++k2)
                            if (!((stash >> k2) & 1ull)) R.put(ch, ninf4);
                    }
                }
            }
            for (int base = tid; walk && base < nch; base += UNR * BLOCK, kb += UNR) {
                uint32_t qv[UNR][4];
                if constexpr (!LDSROW) {       // global working row: batch the v re-reads (L2 / Infinity-Cache hits)
#pragma unroll
                    for (int u = 0; u < UNR; ++u) { const int ch = base + u * BLOCK; if (ch < nch) gload<DT>(p.v, ov, ch, V, p.vec_in, qv[u]); }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int ch = base + u * BLOCK;
                    if (ch >= nch) continue;
                    if constexpr (LDSROW) R.get(ch, qv[u]);
                    bool live = false;
#pragma unroll
                    for (int j = 0; j < EPC; ++j) {
                        float vf = Tr<DT>::to_f(getb<DT>(qv[u], j));
                        live |= !(vf < cutoff) && (ch * EPC + j < V);
                    }
                    if constexpr (PROC) live |= ch == force_ch;
                    if (!live) { if (!sparse) R.put(ch, ninf4); continue; }
                    if (kb + u < 64) livemask |= 1ull << (kb + u);
                    const unsigned slot = atomicAdd(&sm.live_n, 1u);
                    if (slot < (unsigned)LIVE_CAP) {
                        LL.ch[slot] = ch;
                        LL.data[slot] = make_uint4(qv[u][0], qv[u][1], qv[u][2], qv[u][3]);
                        if constexpr (LDSROW) {
                            const int r = ch / BLOCK - p.kl;
                            if (r == 0) rslot[0] = (int)slot; else if (r == 1) rslot[1] = (int)slot; else if (r == 2) rslot[2] = (int)slot;
                        }
                    } else {                   // list full: in place
                        uint32_t qc[4], qd[4], x4[4];
                        gload<DT, true>(p.c, oc, ch, V, p.vec_in, qc);
                        if constexpr (BOTH) gload<DT, true>(p.d, od, ch, V, p.vec_in, qd);
                        contrast_chunk(both_c, ch, qv[u], qc, qd, x4);
                        R.put(ch, x4);
                    }
                }
            }
            __syncthreads();
            const int n_live = (int)min(sm.live_n, (unsigned)LIVE_CAP);
            for (int s0 = tid; s0 < n_live; s0 += BLOCK) {
                const int ch = LL.ch[s0];
                const uint4 vq = LL.data[s0];
                const uint32_t qv[4] = {vq.x, vq.y, vq.z, vq.w};
                uint32_t qc[4], qd[4], x4[4];
                gload<DT, true>(p.c, oc, ch, V, p.vec_in, qc);
                if constexpr (BOTH) gload<DT, true>(p.d, od, ch, V, p.vec_in, qd);
                contrast_chunk(both_c, ch, qv, qc, qd, x4);
                LL.data[s0] = make_uint4(x4[0], x4[1], x4[2], x4[3]);
            }
            __syncthreads();
            for (int s0 = tid; s0 < n_live; s0 += BLOCK) {
                const int ch = LL.ch[s0];
                bool shared_part = true;
                if constexpr (LDSROW) shared_part = ch / BLOCK < p.kl;
                if (shared_part) { const uint4 x = LL.data[s0]; const uint32_t w[4] = {x.x, x.y, x.z, x.w}; R.put(ch, w); }
            }
            if constexpr (LDSROW) {
#pragma unroll
                for (int r = 0; r < NREG; ++r)
                    if (rslot[r] >= 0) { const uint4 x = LL.data[rslot[r]]; rc[r][0] = x.x; rc[r][1] = x.y; rc[r][2] = x.z; rc[r][3] = x.w; }
            }
        };
        if (both) pass_b(std::true_type{});
        else pass_b(std::false_type{});
    } else {
        // ---- plain path (vcd_sample.py:204-205): x = warp(v) ---------------------------
        livemask = ~0ull;
        for (int base = tid; base < nch; base += UNR * BLOCK) {
            uint32_t q[UNR][4];
#pragma unroll
            for (int u = 0; u < UNR; ++u) { const int ch = base + u * BLOCK; if (ch < nch) gload<DT, LDSROW>(p.v, ov, ch, V, p.vec_in, q[u]); }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ch = base + u * BLOCK;
                if (ch < nch) {
#pragma unroll
                    for (int j = 0; j < EPC; ++j) {
                        const int idx = ch * EPC + j;
                        uint32_t b = getb<DT>(q[u], j);
                        float x = Tr<DT>::to_f(b);
                        if (PROC && proc) {                                                   // :204 logits_processor
                            if ((proc & 1) && is_eos(idx)) { x = -INFINITY; b = NINF; }
                            if ((proc & 2) && (long long)idx == p.force_id) { x = rnd<DT>(p.force_val); b = Tr<DT>::from_f(x); }
                        }
                        if (p.use_temp) { x = rnd<DT>(recip ? __fmul_rn(x, p.inv_temp) : __fdiv_rn(x, p.temp)); b = Tr<DT>::from_f(x); }
                        if (idx >= V) b = NINF;
                        setb<DT>(q[u], j, b);
                        if (b != NINF) { nfin += 1; m = fmaxf(m, x); t_nan |= (x != x) ? 1 : 0; t_pinf |= (x == INFINITY) ? 1 : 0; }
                    }
                    R.put(ch, q[u]);
                }
            }
        }
    }
    const float m_thr = m;                                   // this thread's own maximum (the top-k candidate bound below)
    int flags2 = t_nan + 2048 * t_pinf;                      // per-thread 0/1 flags: sums stay < 2^22
    block_max_count2(m, nfin, flags2, sm, lane, wave);
    const bool has_nan = (flags2 % 2048) != 0;
    const bool has_pinf = (flags2 / 2048) != 0;
    const bool row_bad = (nfin == 0) || has_nan || has_pinf;

    auto flagged = [&](int k) { return k >= 64 || ((livemask >> k) & 1ull) != 0ull; };

    // ---- top-k on a row with many candidates (plain sampling: all V of them): by a candidate LIST instead of the radix selection.
    // In every wave take the j-th largest of the 64 per-thread maxima, j = ceil(k / 8); the smallest of those eight values, L, has at
    // least 8 j >= k elements at or above it, so the k-th largest element of the row is among {x >= L} - a few dozen to a few
    // hundred entries for k = 50.  They are gathered into an LDS list, the k-th largest is found by counting, and the row is
    // masked in ONE pass that also recounts the survivors and rebuilds the livemask (usually <= 64 survive: the single-wave tail
    // takes over).  The radix selection it replaces makes two passes over the row with one LDS atomic per element, most of them
    // on a handful of exponent bins (same-address atomics serialise): 175 us of a 357-us launch at B = 4096, V = 32000, k = 50.
    bool topk_done = false;
    if (!row_bad && p.top_k > 0 && nfin > 64 && nch <= 64 * BLOCK) {
        const int k = p.top_k < p.min_keep ? p.min_keep : p.top_k;
        if (k < nfin && k <= BLOCK) {
            uint32_t* ck = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(&sm) + ((sizeof(Smem) + 15) & ~(size_t)15));
            const int jj = (k + NWAVE - 1) / NWAVE;
            int gt = 0;
            for (int q = 0; q < 64; ++q) { const float o = __shfl(m_thr, q); gt += (o > m_thr || (o == m_thr && q < lane)) ? 1 : 0; }
            const float wl = -wave_max((gt == jj - 1) ? -m_thr : -INFINITY);          // the wave's jj-th largest thread maximum
            if (lane == 0) sm.f[0][wave] = wl;
            if (tid == 0) sm.cand_n = 0u;
            __syncthreads();
            float L = sm.f[0][0];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) L = fminf(L, sm.f[0][w]);
            for (int ch = tid, kq = 0; ch < nch; ch += BLOCK, ++kq) {
                if (!flagged(kq)) continue;
                uint32_t w[4]; R.get(ch, w);
                unsigned hit = 0u;                          // branch-free test of the chunk first: nearly every chunk has no candidate
#pragma unroll
                for (int j = 0; j < EPC; ++j) hit |= (Tr<DT>::to_f(getb<DT>(w, j)) >= L ? 1u : 0u) << j;       // -inf >= L is false
                if (hit) {
#pragma unroll
                    for (int j = 0; j < EPC; ++j)
                        if ((hit >> j) & 1u) {
                            const unsigned slot = atomicAdd(&sm.cand_n, 1u);
                            if (slot < (unsigned)TOPK_LIST_MAX) ck[slot] = okey<DT>(getb<DT>(w, j));
                        }
                }
            }
            __syncthreads();
            const int n = (int)sm.cand_n;
            if (n <= TOPK_LIST_MAX) {                   // (n >= k by construction; a row of ties can overflow the list: radix path below)
                for (int e = tid; e < n; e += BLOCK) {
                    const uint32_t key = ck[e];
                    int g = 0, ge = 0;
                    for (int q = 0; q < n; ++q) { const uint32_t o = ck[q]; g += (o > key) ? 1 : 0; ge += (o >= key) ? 1 : 0; }
                    if (g < k && k <= ge) sm.sel[0] = key;     // every entry holding the k-th largest value writes the same key
                }
                __syncthreads();
                const uint32_t kth = sm.sel[0];
                int cnt = 0, zero = 0; unsigned long long lm2 = 0ull;
                for (int ch = tid, kq = 0; ch < nch; ch += BLOCK, ++kq) {
                    if (!flagged(kq)) continue;
                    uint32_t w[4]; R.get(ch, w);
                    int kept = 0;                               // branch-free: almost every chunk ends up all -inf
#pragma unroll
                    for (int j = 0; j < EPC; ++j) {
                        const uint32_t b = getb<DT>(w, j);
                        const bool keep = b != NINF && okey<DT>(b) >= kth;
                        setb<DT>(w, j, keep ? b : NINF);
                        kept += keep ? 1 : 0;
                    }
                    R.put(ch, w);
                    cnt += kept;
                    if (kept && kq < 64) lm2 |= 1ull << kq;
                }
                livemask = lm2;
                float mm = m;
                block_max_count2(mm, cnt, zero, sm, lane, wave);
                nfin = cnt;
                topk_done = true;
            }
            if (tid == 0) sm.cand_n = 0u;               // the single-wave tail gathers into the same counter
            __syncthreads();
        }
    }
    auto store_scores = [&]() {
        if (p.scores != nullptr) {
            if constexpr (LDSROW) {
                for (int ch = tid; ch < nch; ch += BLOCK) { uint32_t w[4]; R.get(ch, w); gstore<DT, true>(p.scores, (long long)row * p.ss, ch, V, p.vec_out, w); }
            } else if (p.scores != p.work) {
                for (int ch = tid; ch < nch; ch += BLOCK) { uint32_t w[4]; R.get(ch, w); gstore<DT>(p.scores, (long long)row * p.ss, ch, V, p.vec_out, w); }
            }
        }
    };
    const bool want_top = p.top_prob != nullptr && p.n_top > 0;

    // ---- single-wave path: rows that kept <= 64 candidates (the usual case after the plausibility mask) -------------
    // The candidates are gathered into a 64-entry LDS list; wave 0 alone applies the warpers (top-k by counting, top-p by
    // mass, same definitions as the block-wide radix selection below), then - while the other 15 waves only store their
    // share of the scores row - sorts the list into the thread-major enumeration order of the general path, scans,
    // draws (or arg-maxes) and extracts the top-n.  One or two barriers instead of three passes over the row and five
    // block-wide barriers, during which the 64 KiB LDS row and all 16 wave slots of the workgroup were held.
    if (!row_bad && nfin <= FAST_TAIL_MAX && nch <= 64 * BLOCK) {
        const bool warp_on = p.top_k > 0 || p.use_topp;
        const bool tail_on = !((p.flags & VDD_NO_SAMPLE) && !want_top);
        if (warp_on || tail_on) {
            const int kmax = (nch + BLOCK - 1) / BLOCK;
            for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
                if (!flagged(k)) continue;
                uint32_t w[4]; R.get(ch, w);
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    const uint32_t b = getb<DT>(w, j);
                    if (b != NINF) {
                        const unsigned slot = atomicAdd(&sm.cand_n, 1u);
                        sm.cand_rank[slot] = (unsigned)((tid * kmax + k) * EPC + j);
                        sm.cand_x[slot] = Tr<DT>::to_f(b);
                        sm.cand_idx[slot] = ch * EPC + j;
                    }
                }
            }
            __syncthreads();
        }
        const int n = nfin;
        const bool act = lane < n;
        unsigned rk = 0xFFFFFFFFu; float xv = -INFINITY; int id = 0x7fffffff;
        bool alive = false;
        if (wave == 0 && (warp_on || tail_on)) {
            if (act) { rk = sm.cand_rank[lane]; xv = sm.cand_x[lane]; id = sm.cand_idx[lane]; }
            alive = act;
        }
        if (warp_on) {
            if (wave == 0) {
                float T = -INFINITY;
                if (p.top_k > 0) {                                   // HF TopKLogitsWarper: scores < k-th largest -> -inf, ties kept
                    const int k = p.top_k < p.min_keep ? p.min_keep : p.top_k;
                    if (k < n) {
                        int ge = 0;
                        for (int q = 0; q < n; ++q) { const float xq = __shfl(xv, q); ge += (xq >= xv) ? 1 : 0; }
                        const float vk = wave_max((act && ge >= k) ? xv : -INFINITY);
                        T = vk;
                        alive = alive && xv >= vk;
                    }
                }
                int Tidx = 0;                                        // tie rule of the exact top-p: (value, index) threshold
                if (p.use_topp && !(p.flags & VDD_TOPP_FP32_MASS)) {
                    // HF TopPLogitsWarper as torch-CPU computes it: probabilities rounded to the model dtype, summed one by one in
                    // ascending (value, index) order in fp32 (fp64 for an fp32 model), every partial sum rounded to the model dtype
                    // and compared with fl(1 - p); the last min_keep of the order are never removed
                    const unsigned long long am = __ballot(alive);
                    const int n_alive = __popcll(am);
                    int pos = 0;
                    for (int q = 0; q < n; ++q) {
                        const float xq = __shfl(xv, q); const int iq = __shfl(id, q);
                        pos += (((am >> q) & 1ull) != 0ull && (xq < xv || (xq == xv && iq < id))) ? 1 : 0;
                    }
                    const float e = alive ? expf(xv - m) : 0.f;
                    const float z = wave_sum(e);
                    if (alive) sm.cand_x[pos] = rnd<DT>(__fdiv_rn(e, z));       // xv / id are in registers by now
                    const float thr = rnd<DT>(p.one_minus_p);
                    int kdrop = 0;
                    if constexpr (DT == VDD_F32) {
                        double acc = 0.0;
                        for (int q = 0; q < n_alive; ++q) { acc += (double)sm.cand_x[q]; if ((float)acc <= thr) kdrop = q + 1; else break; }
                    } else {
                        float acc = 0.f;
                        for (int q = 0; q < n_alive; ++q) { acc += sm.cand_x[q]; if (rnd<DT>(acc) <= thr) kdrop = q + 1; else break; }
                    }
                    const int kd = min(kdrop, max(0, n_alive - p.min_keep));
                    if (kd > 0) {
                        const bool first_kept = alive && pos == kd;
                        const float tv = wave_max(first_kept ? xv : -INFINITY);
                        T = tv;                                      // >= the top-k threshold: top-p works on what top-k kept
                        int ti = first_kept ? id : 0;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) ti = max(ti, __shfl_xor(ti, o));
                        Tidx = ti;
                        alive = alive && pos >= kd;
                    }
                } else if (p.use_topp) {                             // VDD_TOPP_FP32_MASS: cumulative fp32 mass <= fl(1-p) * Z removed, ties together
                    const float e = alive ? __expf(xv - m) : 0.f;
                    const float z = wave_sum(e);
                    const float thr = rnd<DT>(p.one_minus_p) * z;
                    float mass_le = 0.f;
                    for (int q = 0; q < n; ++q) { const float xq = __shfl(xv, q), eq = __shfl(e, q); mass_le += (xq <= xv) ? eq : 0.f; }
                    const bool crossed = alive && mass_le > thr;
                    float pv = crossed ? xv : INFINITY;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) pv = fminf(pv, __shfl_xor(pv, o));
                    float keepv = m;                                 // never remove the top min_keep entries
                    if (p.min_keep > 1) {
                        const int n_alive = __popcll(__ballot(alive));
                        keepv = -INFINITY;
                        if (p.min_keep < n_alive) {
                            int ge = 0;
                            for (int q = 0; q < n; ++q) { const float xq = __shfl(alive ? xv : -INFINITY, q); ge += (xq >= xv) ? 1 : 0; }
                            keepv = wave_max((alive && ge >= p.min_keep) ? xv : -INFINITY);
                        }
                    }
                    const float Tp = (pv < INFINITY) ? fminf(pv, keepv) : keepv;
                    T = fmaxf(T, Tp);
                    alive = alive && xv >= Tp;
                }
                if (lane == 0) { sm.sel[0] = (T == -INFINITY) ? 0u : okey<DT>(Tr<DT>::from_f(T)); sm.sel[1] = (unsigned)Tidx; }
            }
            __syncthreads();
            const uint32_t thr_key = sm.sel[0];
            const int thr_idx = (int)sm.sel[1];
            if (thr_key != 0u) {
                for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
                    if (!flagged(k)) continue;
                    uint32_t w[4]; R.get(ch, w);
                    bool changed = false;
#pragma unroll
                    for (int j = 0; j < EPC; ++j) {
                        const uint32_t b = getb<DT>(w, j);
                        if (b == NINF) continue;
                        const uint32_t kk = okey<DT>(b);
                        if (kk < thr_key || (kk == thr_key && ch * EPC + j < thr_idx)) { setb<DT>(w, j, NINF); changed = true; }
                    }
                    if (changed) R.put(ch, w);
                }
            }
        }
        if (tid == 0 && p.status) p.status[row] = VDD_ROW_OK;
        store_scores();
        if (wave != 0 || !tail_on) return;
        if (!alive) xv = -INFINITY;
        // numerators in enumeration order (sorted through LDS: every lane has its entry in registers by now)
        {
            int pos = 0;
            for (int q = 0; q < n; ++q) { const unsigned rq = __shfl(rk, q); pos += (rq < rk) ? 1 : 0; }
            if (act) { sm.cand_x[pos] = alive ? __expf(xv - m) : 0.f; sm.cand_idx[pos] = id; }
        }
        const float e_s = act ? sm.cand_x[lane] : 0.f;           // same wave: LDS program order
        const int id_s = act ? sm.cand_idx[lane] : -1;
        float incl = e_s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float nb = __shfl_up(incl, o); if (lane >= o) incl += nb; }
        const float zsum = __shfl(incl, 63);
        if (want_top) {                                              // value descending, index ascending
            float pv = INFINITY; int pi = -1;
            for (int r = 0; r < p.n_top; ++r) {
                const bool after = alive && (xv < pv || (xv == pv && id > pi));
                float bv = after ? xv : -INFINITY; int bi = after ? id : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov2 = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                    if (ov2 > bv || (ov2 == bv && oi < bi)) { bv = ov2; bi = oi; }
                }
                const bool have = bi != 0x7fffffff;
                if (lane == 0) {
                    p.top_prob[(long long)row * p.n_top + r] = have ? rnd<DT>(__fdiv_rn(__expf(bv - m), zsum)) : 0.f;
                    p.top_tok[(long long)row * p.n_top + r] = have ? bi : -1;
                }
                pv = have ? bv : -INFINITY; pi = bi;
            }
        }
        if (p.flags & VDD_NO_SAMPLE) return;
        int tok_i;
        if (p.flags & VDD_PICK_ARGMAX) {
            float bv = xv; int bi = alive ? id : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov2 = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                if (ov2 > bv || (ov2 == bv && oi < bi)) { bv = ov2; bi = oi; }
            }
            tok_i = bi;
        } else {
            const float u = p.uniforms ? p.uniforms[row] : philox_uniform(p.seed, p.offset + (p.offset_ptr ? *p.offset_ptr : 0ull), (unsigned)row);
            const float target = u * zsum;
            const unsigned long long hits = __ballot(act && e_s > 0.f && target < incl);
            const unsigned long long mass = __ballot(act && e_s > 0.f);
            const int pl = hits ? (int)__builtin_ctzll(hits) : (mass ? 63 - (int)__clzll((long long)mass) : n - 1);   // rounding: last entry holding mass
            tok_i = __shfl(id_s, pl);
        }
        if (lane == 0) {
            long long tok = (long long)tok_i;
            if (p.unfinished != nullptr && p.n_eos > 0) {
                long long uf = p.unfinished[row];
                tok = tok * uf + p.pad * (1 - uf);                                              // :260
                long long keep = 1;
                for (int e = 0; e < p.n_eos; ++e) keep *= (tok != p.eos[e]) ? 1 : 0;             // :286-288
                p.unfinished[row] = uf * keep;
            }
            p.next_tokens[(long long)row * p.st] = tok;
        }
        return;
    }

    // ---- top-k (HF TopKLogitsWarper: scores < kth -> -inf, ties kept) ------------
    if (p.top_k > 0 && !has_nan && nfin > 0 && !topk_done) {
        int k = p.top_k < p.min_keep ? p.min_keep : p.top_k;
        if (k < nfin) {
            uint32_t kth = select_kth_key<DT, LDSROW>(R, nch, (unsigned)k, sm, tid, lane, wave, livemask);
            mask_below_key<DT, LDSROW>(R, nch, kth, tid, livemask);
        }
    }

    // ---- top-p (HF TopPLogitsWarper) ----------------------------------------------
    bool topp_done = false;
    if (p.use_topp && !row_bad && !(p.flags & VDD_TOPP_FP32_MASS) &&
        (p.top_k > 0 ? (p.top_k < p.min_keep ? p.min_keep : p.top_k) <= TOPP_EXACT_MAX : nfin <= TOPP_EXACT_MAX)) {
        // exact form on an explicit candidate list (what survived top-k), same arithmetic as the single-wave path
        ToppScratch& ts = *reinterpret_cast<ToppScratch*>(reinterpret_cast<unsigned char*>(&sm) + ((sizeof(Smem) + 15) & ~(size_t)15));
        if (tid == 0) sm.cand_n = 0u;
        __syncthreads();
        for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
            if (!flagged(k)) continue;
            uint32_t w[4]; R.get(ch, w);
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const uint32_t b = getb<DT>(w, j);
                if (b != NINF) {
                    const unsigned slot = atomicAdd(&sm.cand_n, 1u);
                    if (slot < (unsigned)TOPP_EXACT_MAX) { ts.gx[slot] = Tr<DT>::to_f(b); ts.gi[slot] = ch * EPC + j; }
                }
            }
        }
        __syncthreads();
        const int n = (int)sm.cand_n;
        if (n <= TOPP_EXACT_MAX) {                                  // (ties at the top-k threshold can push a row past the bound)
            float x2[2]; int i2[2], pos2[2]; float e2[2];
            float zp = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = tid + u * BLOCK;
                x2[u] = c < n ? ts.gx[c] : INFINITY; i2[u] = c < n ? ts.gi[c] : 0x7fffffff; pos2[u] = 0;
                e2[u] = c < n ? expf(x2[u] - m) : 0.f;
                zp += e2[u];
            }
            for (int q = 0; q < n; ++q) {
                const float xq = ts.gx[q]; const int iq = ts.gi[q];
#pragma unroll
                for (int u = 0; u < 2; ++u) pos2[u] += (xq < x2[u] || (xq == x2[u] && iq < i2[u])) ? 1 : 0;
            }
            const float z = block_sum(zp, sm, lane, wave);
#pragma unroll
            for (int u = 0; u < 2; ++u) if (tid + u * BLOCK < n) ts.sp[pos2[u]] = rnd<DT>(__fdiv_rn(e2[u], z));
            __syncthreads();
            if (tid == 0) {
                const float thr = rnd<DT>(p.one_minus_p);
                int kdrop = 0;
                if constexpr (DT == VDD_F32) {
                    double acc = 0.0;
                    for (int q = 0; q < n; ++q) { acc += (double)ts.sp[q]; if ((float)acc <= thr) kdrop = q + 1; else break; }
                } else {
                    float acc = 0.f;
                    for (int q = 0; q < n; ++q) { acc += ts.sp[q]; if (rnd<DT>(acc) <= thr) kdrop = q + 1; else break; }
                }
                sm.sel[2] = (unsigned)min(kdrop, max(0, n - p.min_keep));
                sm.sel[0] = 0u; sm.sel[1] = 0u;
            }
            __syncthreads();
            const int kd = (int)sm.sel[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (kd > 0 && tid + u * BLOCK < n && pos2[u] == kd) { sm.sel[0] = okey<DT>(Tr<DT>::from_f(x2[u])); sm.sel[1] = (unsigned)i2[u]; }
            __syncthreads();
            if (kd > 0) mask_below_key<DT, LDSROW>(R, nch, sm.sel[0], tid, livemask, (int)sm.sel[1]);
            topp_done = true;
        }
    }
    if (p.use_topp && !row_bad && !topp_done) {                     // more candidates than the list holds (or VDD_TOPP_FP32_MASS):
        float z = 0.f, zb = 0.f;                            // total mass; mass of everything below the row maximum
        for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
            if (!(k >= 64 || ((livemask >> k) & 1ull) != 0ull)) continue;
            uint32_t w[4]; R.get(ch, w);
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                uint32_t b = getb<DT>(w, j);
                if (b != NINF) { const float x = Tr<DT>::to_f(b), e = __expf(x - m); z += e; zb += (x < m) ? e : 0.f; }
            }
        }
        z = block_sum(z, sm, lane, wave);
        zb = block_sum(zb, sm, lane, wave);
        const float thr = rnd<DT>(p.one_minus_p) * z;      // cum <= fl(1-p)  <=>  mass <= fl(1-p) * Z
        constexpr int TOPD = 4, KB = Tr<DT>::KEYBITS;
        const uint32_t mkey = okey<DT>(Tr<DT>::from_f(m)), mdig = mkey >> (KB - 8);
        uint32_t pkey = 0;
        bool crossed;
        // a peaked row (the usual one when a model is confident): everything below the maximum together is mass top-p removes, so
        // the threshold is the maximum itself and the two radix passes over the row are not needed
        if (zb <= thr) { crossed = true; pkey = mkey; }
        else {
            // The TOP DIGIT of the threshold key (sign + 7 exponent bits for bf16 / fp32: two octaves per digit) in registers: a
            // row puts almost all of its elements into a handful of top digits, so the shared-memory histogram of that digit is
            // thousands of same-address atomics per wave (956 us at B = 4096 on flat rows); the mass of the maximum's digit and
            // of the TOPD - 1 under it is five selects per element.
            float zd[TOPD] = {0.f, 0.f, 0.f, 0.f}, zrest = 0.f;
            for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
                if (!(k >= 64 || ((livemask >> k) & 1ull) != 0ull)) continue;
                uint32_t w[4]; R.get(ch, w);
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    uint32_t b = getb<DT>(w, j);
                    if (b != NINF) {
                        const float e = __expf(Tr<DT>::to_f(b) - m);
                        const uint32_t dd = mdig - (okey<DT>(b) >> (KB - 8));
#pragma unroll
                        for (int q = 0; q < TOPD; ++q) zd[q] += (dd == (uint32_t)q) ? e : 0.f;
                        zrest += (dd >= (uint32_t)TOPD) ? e : 0.f;
                    }
                }
            }
            zrest = block_sum(zrest, sm, lane, wave);
#pragma unroll
            for (int q = 0; q < TOPD; ++q) zd[q] = block_sum(zd[q], sm, lane, wave);
            if (zrest > thr) {
                // the threshold lies more than TOPD digits under the maximum: full histogram passes
                crossed = select_mass_key<DT, LDSROW>(R, nch, m, thr, sm, tid, lane, wave, pkey, livemask);
            } else {
                float below = zrest; int dq = -1;
#pragma unroll
                for (int q = TOPD - 1; q >= 0; --q)
                    if (dq < 0) { if (below + zd[q] > thr) dq = q; else below += zd[q]; }
                if (dq < 0) {             // rounding: zb > thr but the digit sums did not cross - the maximum's own digit holds it
                    dq = 0; below -= zd[0];
                }
                crossed = select_mass_key<DT, LDSROW>(R, nch, m, thr, sm, tid, lane, wave, pkey, livemask, 1,
                                                      (mdig - (uint32_t)dq) << (KB - 8), below);
            }
        }
        // never remove the top min_keep entries
        uint32_t keep_key;
        if (p.min_keep <= 1) keep_key = okey<DT>(Tr<DT>::from_f(m));
        else keep_key = (p.min_keep < nfin) ? select_kth_key<DT, LDSROW>(R, nch, (unsigned)p.min_keep, sm, tid, lane, wave, livemask) : 0u;
        uint32_t thr_key = crossed ? (pkey < keep_key ? pkey : keep_key) : keep_key;
        mask_below_key<DT, LDSROW>(R, nch, thr_key, tid, livemask);
    }

    // ---- scores row out: issued LAST (after the token), see the end of the kernel ----------------
    if (row_bad) {
        store_scores();
        if (tid == 0) {
            if (p.status) p.status[row] = VDD_ROW_EMPTY;
            if (p.next_tokens && !(p.flags & VDD_NO_SAMPLE)) p.next_tokens[(long long)row * p.st] = -1;
        }
        if (p.top_prob && tid < p.n_top) { p.top_prob[(long long)row * p.n_top + tid] = 0.f; p.top_tok[(long long)row * p.n_top + tid] = -1; }
        return;
    }
    if (tid == 0 && p.status) p.status[row] = VDD_ROW_OK;
    store_scores();          // the row is final: its stores drain while the tail below runs on LDS / registers
    if ((p.flags & VDD_NO_SAMPLE) && !want_top) return;

    // ---- per-thread mass, block scan (thread-major order) ---------------------------

    float t = 0.f;
    for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
        if (!flagged(k)) continue;
        uint32_t w[4]; R.get(ch, w);
#pragma unroll
        for (int j = 0; j < EPC; ++j) { uint32_t b = getb<DT>(w, j); if (b != NINF) t += __expf(Tr<DT>::to_f(b) - m); }
    }
    float incl = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float n = __shfl_up(incl, o); if (lane >= o) incl += n; }
    if (lane == 63) sm.f[0][wave] = incl;
    __syncthreads();
    float base = 0.f, Z = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) { if (w == wave) base = Z; Z += sm.f[0][w]; }
    incl += base;
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = base;
    __syncthreads();

    // ---- top-n of softmax(scores) (metrics.py:103-104) -------------------------------
    if (want_top) {
        float pv = INFINITY; int pi = -1;     // previous pick (value desc, index asc)
        for (int r = 0; r < p.n_top; ++r) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
                if (!flagged(k)) continue;
                uint32_t w[4]; R.get(ch, w);
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    uint32_t b = getb<DT>(w, j);
                    const int idx = ch * EPC + j;
                    float f = Tr<DT>::to_f(b);
                    bool after = (b != NINF) && (f < pv || (f == pv && idx > pi));
                    if (after && (f > bv || (f == bv && idx < bi))) { bv = f; bi = idx; }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                float ov2 = __shfl_xor(bv, o); int oi = __shfl_xor(bi, o);
                if (ov2 > bv || (ov2 == bv && oi < bi)) { bv = ov2; bi = oi; }
            }
            if (lane == 0) { sm.f[1][wave] = bv; sm.i[1][wave] = bi; }
            __syncthreads();
            bv = sm.f[1][0]; bi = sm.i[1][0];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) {
                float ov2 = sm.f[1][w]; int oi = sm.i[1][w];
                if (ov2 > bv || (ov2 == bv && oi < bi)) { bv = ov2; bi = oi; }
            }
            __syncthreads();
            const bool have = bi != 0x7fffffff;
            if (tid == 0) {
                p.top_prob[(long long)row * p.n_top + r] = have ? rnd<DT>(__fdiv_rn(__expf(bv - m), Z)) : 0.f;
                p.top_tok[(long long)row * p.n_top + r] = have ? bi : -1;
            }
            pv = have ? bv : -INFINITY; pi = bi;
        }
    }
    if (p.flags & VDD_NO_SAMPLE) return;

    // ---- token: argmax or inverse-CDF draw in thread-major order ---------------------
    if (tid == 0) sm.sel[3] = 0xFFFFFFFFu;
    __syncthreads();
    if (p.flags & VDD_PICK_ARGMAX) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
            if (!flagged(k)) continue;
            uint32_t w[4]; R.get(ch, w);
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                uint32_t b = getb<DT>(w, j);
                const int idx = ch * EPC + j;
                float f = Tr<DT>::to_f(b);
                if (b != NINF && (f > bv || (f == bv && idx < bi))) { bv = f; bi = idx; }
            }
        }
        if (bv == m) atomicMin(&sm.sel[3], (unsigned)bi);
    } else {
        const float u = p.uniforms ? p.uniforms[row] : philox_uniform(p.seed, p.offset + (p.offset_ptr ? *p.offset_ptr : 0ull), (unsigned)row);
        const float target = u * Z;
        bool hit = (t > 0.f) && (target >= excl) && (target < incl);
        // rounding fallback: target >= Z lands on the last thread holding mass
        unsigned long long anyhit = __ballot(hit);
        unsigned long long mass = __ballot(t > 0.f);
        if (lane == 0) { sm.i[0][wave] = anyhit ? 1 : 0; sm.i[1][wave] = mass ? (wave * 64 + 63 - __clzll((long long)mass)) : -1; }
        __syncthreads();
        int any = 0, last = -1;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) { any |= sm.i[0][w]; last = sm.i[1][w] > last ? sm.i[1][w] : last; }
        if (!any) hit = (tid == last);
        if (hit) {
            float acc = excl; int pick = -1, lastfin = -1;
            for (int ch = tid, k = 0; ch < nch; ch += BLOCK, ++k) {
                if (!flagged(k)) continue;
                uint32_t w[4]; R.get(ch, w);
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    uint32_t b = getb<DT>(w, j);
                    const int idx = ch * EPC + j;
                    if (b != NINF) {
                        acc += __expf(Tr<DT>::to_f(b) - m);
                        lastfin = idx;
                        if (pick < 0 && target < acc) pick = idx;
                    }
                }
            }
            sm.sel[3] = (unsigned)(pick >= 0 ? pick : lastfin);
        }
    }
    __syncthreads();
    if (tid == 0) {
        long long tok = (long long)sm.sel[3];
        if (p.unfinished != nullptr && p.n_eos > 0) {
            long long uf = p.unfinished[row];
            tok = tok * uf + p.pad * (1 - uf);                                              // :260
            long long keep = 1;
            for (int e = 0; e < p.n_eos; ++e) keep *= (tok != p.eos[e]) ? 1 : 0;             // :286-288
            p.unfinished[row] = uf * keep;
        }
        p.next_tokens[(long long)row * p.st] = tok;
    }
}

// ------------------------------------------------------------------ host side
thread_local char g_err[256] = "";

int fail(int code, const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }

template <int DT, bool L, bool PROC>
int launch_one(const KP& kp, size_t lds, hipStream_t st) {
    static int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&vdd_contrast_sample_kernel<DT, L, PROC>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr_rc;
    hipLaunchKernelGGL((vdd_contrast_sample_kernel<DT, L, PROC>), dim3(kp.B), dim3(BLOCK), lds, st, kp);
    return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH;
}

template <int DT>
int launch_dt(const KP& kp, bool ldsrow, size_t lds, hipStream_t st) {
    const bool proc = kp.eos_min != nullptr || kp.force != nullptr;
    if (proc) return ldsrow ? launch_one<DT, true, true>(kp, lds, st) : launch_one<DT, false, true>(kp, lds, st);
    return ldsrow ? launch_one<DT, true, false>(kp, lds, st) : launch_one<DT, false, false>(kp, lds, st);
}

size_t esize(int dtype) { return dtype == VDD_F32 ? 4 : 2; }

}  // namespace

extern "C" {

int vdd_abi_version(void) { return VDD_ABI_VERSION; }
const char* vdd_last_error(void) { return g_err; }
// Largest V whose working row stays on chip: LDS part (<= LDS_ROW_BYTES_MAX) + NREG register chunks per thread.
int vdd_lds_row_capacity(int dtype) {
    const int epc = (int)(16 / esize(dtype));
    const int kl_max = LDS_ROW_BYTES_MAX / (BLOCK * 16);
    return (kl_max + NREG) * BLOCK * epc;
}

int vdd_topp_exact_max(void) { return TOPP_EXACT_MAX; }

const char* vdd_kernel_name(int dtype, int V) {
    (void)dtype; (void)V;
    return "vdd_contrast_sample_kernel";
}

int vdd_contrast_sample(const vdd_sample_params* p, void* hip_stream) {
    if (!p) return fail(VDD_ERR_INVALID_ARG, "params is NULL");
    if (p->abi_version != VDD_ABI_VERSION) return fail(VDD_ERR_INVALID_ARG, "abi_version mismatch");
    if (p->B < 0 || p->V <= 0) return fail(VDD_ERR_INVALID_ARG, "B < 0 or V <= 0");
    if (p->B == 0) return VDD_OK;
    if (!p->logit_v) return fail(VDD_ERR_INVALID_ARG, "logit_v is NULL");
    if (p->logit_dd && !p->logit_cd) return fail(VDD_ERR_INVALID_ARG, "logit_dd given without logit_cd");
    if (!(p->flags & VDD_NO_SAMPLE) && !p->next_tokens) return fail(VDD_ERR_INVALID_ARG, "next_tokens is NULL");
    if ((p->flags & VDD_NO_SAMPLE) && !p->scores_out && !p->top_prob) return fail(VDD_ERR_INVALID_ARG, "VDD_NO_SAMPLE without any output");
    if (p->dtype != VDD_F32 && p->dtype != VDD_F16 && p->dtype != VDD_BF16) return fail(VDD_ERR_INVALID_ARG, "bad dtype");
    if (p->min_keep < 1) return fail(VDD_ERR_INVALID_ARG, "min_keep < 1");
    if (p->n_top < 0 || p->n_top > 16) return fail(VDD_ERR_INVALID_ARG, "n_top out of [0,16]");
    if ((p->top_prob == nullptr) != (p->top_tok == nullptr)) return fail(VDD_ERR_INVALID_ARG, "top_prob/top_tok must be given together");
    if (p->n_eos < 0 || (p->n_eos > 0 && !p->eos_ids)) return fail(VDD_ERR_INVALID_ARG, "eos_ids missing");
    if (p->temperature != p->temperature) return fail(VDD_ERR_INVALID_ARG, "temperature is NaN");
    if (p->eos_min_step && p->n_eos <= 0) return fail(VDD_ERR_INVALID_ARG, "eos_min_step given without eos_ids");
    if (p->force_eos && (p->force_eos_id < 0 || p->force_eos_id >= p->V)) return fail(VDD_ERR_INVALID_ARG, "force_eos_id outside [0, V)");

    const size_t es = esize(p->dtype);
    const int epc = (int)(16 / es);
    const int nch_h = (p->V + epc - 1) / epc;
    const int kmax_h = (nch_h + BLOCK - 1) / BLOCK;                 // chunks per thread
    const int nreg_h = kmax_h < NREG ? kmax_h : NREG;               // the LAST nreg chunks of every thread live in registers
    const int kl_h = kmax_h - nreg_h;
    const size_t row_bytes = (size_t)(nch_h < kl_h * BLOCK ? nch_h : kl_h * BLOCK) * 16;   // LDS part of the row
    const bool ldsrow = row_bytes <= (size_t)LDS_ROW_BYTES_MAX;
    auto al = [&](const void* ptr, long long stride) { return ptr == nullptr || ((((uintptr_t)ptr) & 15u) == 0 && ((stride * (long long)es) & 15) == 0); };
    KP kp{};
    kp.v = p->logit_v; kp.c = p->logit_cd; kp.d = p->logit_dd;
    kp.sv = p->stride_v; kp.sc = p->stride_cd; kp.sd = p->stride_dd; kp.ss = p->stride_scores;
    kp.B = p->B; kp.V = p->V; kp.flags = p->flags;
    kp.kl = ldsrow ? kl_h : (1 << 20);
    kp.min_keep = p->min_keep; kp.top_k = p->top_k > 0 ? p->top_k : 0; kp.n_eos = p->n_eos;
    kp.n_top = p->top_prob ? p->n_top : 0;
    kp.s1 = (float)(1.0 + p->alpha); kp.s2 = (float)p->alpha; kp.log_beta = (float)p->log_beta;
    kp.use_temp = (p->temperature > 0.0 && p->temperature != 1.0) ? 1 : 0;
    kp.temp = kp.use_temp ? (float)p->temperature : 1.0f;
    kp.inv_temp = 1.0f / kp.temp;
    kp.use_topp = (p->top_p >= 0.0 && p->top_p < 1.0) ? 1 : 0;
    kp.one_minus_p = (float)(1.0 - p->top_p);
    kp.seed = p->philox_seed; kp.offset = p->philox_offset; kp.uniforms = p->uniforms;
    kp.offset_ptr = (const unsigned long long*)p->philox_offset_ptr;
    kp.eos = (const long long*)p->eos_ids; kp.pad = p->pad_id;
    kp.unfinished = (long long*)p->unfinished; kp.next_tokens = (long long*)p->next_tokens;
    kp.st = p->stride_tokens > 0 ? p->stride_tokens : 1;
    kp.scores = p->scores_out; kp.top_prob = p->top_prob; kp.top_tok = (long long*)p->top_tok;
    kp.status = p->row_status;
    kp.eos_min = p->eos_min_step; kp.step = p->step; kp.step_ptr = (const long long*)p->step_ptr;
    kp.force = p->force_eos; kp.force_id = p->force_eos_id; kp.force_val = (float)p->force_eos_value;
    kp.vec_in = al(p->logit_v, p->stride_v) && al(p->logit_cd, p->stride_cd) && al(p->logit_dd, p->stride_dd);
    kp.vec_out = al(p->scores_out, p->stride_scores);
    if (!ldsrow) {
        if (p->scores_out) { kp.work = p->scores_out; kp.sw = p->stride_scores; }
        else if (p->workspace) { kp.work = p->workspace; kp.sw = p->stride_workspace; }
        else return fail(VDD_ERR_INVALID_ARG, "V exceeds vdd_lds_row_capacity(dtype): pass scores_out or workspace [B,V]");
        kp.vec_work = al(kp.work, kp.sw);
    }
    const size_t scratch = (kp.c == nullptr && kp.top_k == 0) ? 0 : sizeof(LiveList);     // pass B's work list / the top-k candidate list; the scratch users never overlap in time
    const size_t lds = ((sizeof(Smem) + 15) & ~(size_t)15) + (kp.use_topp && sizeof(ToppScratch) > scratch ? sizeof(ToppScratch) : scratch) + (ldsrow ? row_bytes : 0);
    hipStream_t st = (hipStream_t)hip_stream;
    int rc;
    switch (p->dtype) {
        case VDD_F16: rc = launch_dt<VDD_F16>(kp, ldsrow, lds, st); break;
        case VDD_BF16: rc = launch_dt<VDD_BF16>(kp, ldsrow, lds, st); break;
        default: rc = launch_dt<VDD_F32>(kp, ldsrow, lds, st); break;
    }
    if (rc == VDD_ERR_LAUNCH) return fail(rc, "hipLaunchKernel failed");
    return rc;
}

}  // extern "C"
