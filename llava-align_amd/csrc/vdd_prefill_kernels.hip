// Prefill-side kernels for gfx950: flash-style MFMA attention for the LLM prompt and the
// CLIP ViT, LayerNorm, and bias + activation.  They replace the eager HF modules the
// reference runs at step 0 of every branch (experiments/llava/model/llava_arch.py:82-204 ->
// CLIPVisionModel / LlamaModel [ext]; clip_encoder.py:39-51; multimodal_projector/builder.py:33-46).
//
// vdd_flash_attention: one 256-thread block = 128 query rows of one (sequence, head), 32 per wave; 64-key K / V tiles shared by
// the block through LDS (LDS-DMA, double-buffered), 32x32x16 bf16 MFMAs for S^T = K Q^T and O^T = V^T P^T, softmax and
// rescale lane-local, V read transposed with ds_read_b64_tr_b16 (see flash_attn2_kernel).  Keys come from
// [prefix slot | own slot] like the decode kernel.  Bound: MFMA in the long-sequence limit (0.9 PF/s at 5,120 keys), block
// start-up and causal tile waste at the bench's 611-token prefixes (0.37 PF/s of useful flops).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vdd_elem.h"

namespace {
namespace VDD_ELEM_NS {
using namespace vdd_elem;

typedef __attribute__((ext_vector_type(8))) short frag8_t;    // MFMA A/B fragment: 8 elements (bf16 or fp16 bit patterns)

struct SeqDesc { int q_row0, Tq, pos0, slot, pslot, plen; };

constexpr float NEG_BIG = -1.0e30f;

// ------------------------------------------------------------------ prefill attention, 128-query blocks
// 4 waves x 32 query rows, 64-key tiles, 32x32x16 MFMA, K and V tiles shared by the block through LDS:
//   * K / V tiles go HBM -> LDS by LDS-DMA (global_load_lds, per-lane 64-bit source addresses: a tile may straddle the prefix
//     and the own pool), two tiles ahead of nothing: tile t+1 is issued right behind the barrier that opens tile t (double buffer);
//   * K tile [64 keys][D] row-major, 16-byte chunks XOR-swizzled on the SOURCE side (the DMA image is lane-linear) so that the
//     ds_read_b128 of a K fragment (32 keys x one chunk) is conflict-free;
//   * S^T = K Q^T (A = K fragment, B = Q fragment held in registers): lane (query q = lane & 31, hi) ends up with the scores of
//     its query for keys (r & 3) + 8 (r >> 2) + 4 hi of each 32-key block: max / sum are 32 local values + ONE exchange with
//     lane ^ 32, and the 8 scores r = 8 j .. 8 j + 7 are exactly the k-slice of a 16-key MFMA step: P never leaves registers;
//   * O^T = V^T P^T (A = V^T fragment, B = P fragment): the accumulator column is the lane's OWN query, so the online-softmax
//     rescale and the final normalisation are lane-local (no shuffles);
//   * V tile as 1-KiB [32 keys][16 dims] subtiles (1152 bytes apart: the two dim-halves of a 32-dim MFMA tile land on different
//     banks), read TRANSPOSED with ds_read_b64_tr_b16: within a 16-lane group lane i receives element i % 4 of the 8 bytes
//     addressed by lane 4 j + i / 4 (tools/probes/tr_read_probe.hip), i.e. 4 consecutive keys of one dim from a row-major image.
// One barrier per tile.  Same sequence descriptors and prefix indirection as before.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) void* g_ptr_t;

template <int D, bool CAUSAL, bool PACK>
__global__ void __launch_bounds__(256, 2) flash_attn2_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                                                             const uint16_t* __restrict__ vc, const uint16_t* __restrict__ kpre,
                                                             const uint16_t* __restrict__ vpre, const SeqDesc* __restrict__ seqs,
                                                             uint16_t* __restrict__ out, int H, int Hkv, long long slot_stride,
                                                             int t_max, long long pre_stride, int pre_tmax, float scale, int nx, int n_seq,
                                                             const int4* __restrict__ packs) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!PACK || CAUSAL, "packed mode is the causal suffix pass");
    constexpr int KS = D / 16;                 // 16-deep k-steps of S^T per 32-key block
    constexpr int NT = D / 32;                 // 32-dim tiles of O^T
    constexpr int KROW = D * 2;                // bytes of a K row
    constexpr int KTILE = 64 * KROW;           // K tile bytes
    constexpr int NKI = KTILE / 1024;          // 1-KiB DMA images of the K tile
    constexpr int KPI = 1024 / KROW;           // keys per K image (4 at D = 128, 8 at D = 64)
    constexpr int CPR = KROW / 16;             // 16-byte chunks per K row
    constexpr int DSUB = D / 16;               // 16-dim V subtiles per 32-key half
    constexpr int NVI = 2 * DSUB;              // V subtiles (= DMA images) per tile
    constexpr int VSUB = 1152;                 // bytes between V subtiles (the tr-read offsets below are written out for this value)
    static_assert(NT == 4 || NT == 2, "");
    constexpr int BUF = KTILE + NVI * VSUB;    // one tile buffer
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8, and the NX query blocks of one (sequence, head) read the same K / V, so
    // they take CONSECUTIVE slots of ONE XCD (its L2 then serves all but the first read of a tile; spread round-robin over the
    // XCDs the causal prefix pass pulled every K / V byte 3 times from HBM / Infinity Cache).  Long (late) query blocks first.
    const int NX = nx, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int pair = (local / NX) * 8 + xcd;
    if (pair >= H * n_seq) return;                            // n_seq: sequences, or packs in packed mode
    const int head = pair % H, kvh = head / (H / Hkv);
    // PACKED mode (the suffix pass: ~25 query rows per sequence, one wave's worth): a block takes up to FOUR sequences that
    // continue the same prefix (one per wave; the host lists them in packs[]).  The 64-key tiles lying wholly inside the prefix
    // are staged ONCE and used by all four waves; the rest of each sequence's keys (prefix tail + own tokens) go through the
    // same buffers one sequence after the other, computed by its wave alone.  Unpacked, every sequence staged all its tiles
    // for one active wave in four: 7.8 GB of L2 -> LDS traffic per layer call at 768 questions, which is what bounded the pass.
    int mine = pair / H;                                      // this wave's sequence
    int pk[4] = {-1, -1, -1, -1};
    if constexpr (PACK) {
        const int4 p4 = packs[pair / H];
        pk[0] = p4.x; pk[1] = p4.y; pk[2] = p4.z; pk[3] = p4.w;
        mine = wave == 0 ? p4.x : (wave == 1 ? p4.y : (wave == 2 ? p4.z : p4.w));
    }
    const bool have_seq = mine >= 0;
    const SeqDesc sd = seqs[have_seq ? mine : pk[0]];
    const int qt0 = PACK ? 0 : (NX - 1 - local % NX) * 128;
    if (qt0 >= sd.Tq && !PACK) return;
    const int r0 = PACK ? 0 : qt0 + wave * 32;
    const int Tk = sd.pos0 + sd.Tq;
    const int last_row = min(qt0 + 127, sd.Tq - 1);
    const int kend = CAUSAL ? min(Tk, sd.pos0 + last_row + 1) : Tk;          // block-uniform key bound (unpacked mode)
    const bool wave_has_rows = have_seq && r0 < sd.Tq;
    const int wave_last_pos = sd.pos0 + min(r0 + 31, sd.Tq - 1);

    // Q fragments (B operand of S^T): lane (q, hi) holds Q[r0 + q][16 ks + 8 hi .. + 7]
    frag8_t qf[KS];
    {
        int qr = r0 + ql; if (qr >= sd.Tq) qr = sd.Tq - 1;
        const uint16_t* qp = q + ((size_t)(sd.q_row0 + qr) * H + head) * D + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag8_t*>(qp + ks * 16);
    }
    const int qpos = sd.pos0 + r0 + ql;
    const size_t head_off = (size_t)kvh * t_max * D, pre_off = (size_t)kvh * pre_tmax * D;
    const uint16_t* kbase_pre = kpre + (size_t)sd.pslot * pre_stride + pre_off;
    const uint16_t* vbase_pre = vpre + (size_t)sd.pslot * pre_stride + pre_off;

    // ---- LDS-DMA of one tile of the sequence in own-pool slot `slot_o` with `tk_o` keys (prefix = this block's): (NKI + NVI)
    // 1-KiB images, (NKI + NVI) / 4 per wave; a key past the sequence is clamped (its score is masked)
    auto stage = [&](int kt, int buf, int slot_o, int tk_o) {
        char* base = lds + buf * BUF;
        const uint16_t* kbase_own = kc + (size_t)slot_o * slot_stride + head_off - (size_t)sd.plen * D;
        const uint16_t* vbase_own = vc + (size_t)slot_o * slot_stride + head_off - (size_t)sd.plen * D;
#pragma unroll
        for (int m0 = 0; m0 < NKI / 4; ++m0) {
            const int m = m0 * 4 + wave;
            const int key = m * KPI + lane / CPR, slot = lane % CPR;
            const int c = D == 128 ? (slot ^ (key & 15)) : (slot ^ ((key >> 1) & 7));
            int t = kt + key; if (t >= tk_o) t = tk_o - 1;
            const uint16_t* src = (t < sd.plen ? kbase_pre : kbase_own) + (size_t)t * D + c * 8;
            __builtin_amdgcn_global_load_lds((g_ptr_t)src, (lds_ptr_t)(base + m * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int m0 = 0; m0 < NVI / 4; ++m0) {
            const int sI = m0 * 4 + wave, kh = sI / DSUB, ds = sI % DSUB;
            const int key = 32 * kh + lane / 2, d = 16 * ds + 8 * (lane & 1);
            int t = kt + key; if (t >= tk_o) t = tk_o - 1;
            const uint16_t* src = (t < sd.plen ? vbase_pre : vbase_own) + (size_t)t * D + d;
            __builtin_amdgcn_global_load_lds((g_ptr_t)src, (lds_ptr_t)(base + KTILE + sI * VSUB), 16, 0, 0);
        }
    };

    f32x16_t o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[nt][e] = 0.f;
    float mq = NEG_BIG, lq = 0.f;              // running max (scaled, log2 domain) of query q; partial sum of this lane's keys
    const float sc2 = scale * 1.44269504088896340736f;

    // fragment read offsets (bytes inside a tile buffer)
    const int krow = D == 128 ? (ql * KROW) : (ql * KROW);
    const int kswz = D == 128 ? (ql & 15) : ((ql >> 1) & 7);
    const int vrd = ((lane >> 4) & 1) * VSUB + (4 * hi + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;      // + (b * DSUB + 2 nt) * VSUB + (16 j + 8 h) * 32

    auto compute = [&](int kt, int buf) {                     // this wave's 32 query rows against the tile in `buf`
        const char* kb = lds + buf * BUF;
        const char* vb = kb + KTILE;
        // ---- S^T = K Q^T: all 2 KS fragment reads in flight, then the two independent accumulator chains interleaved ----
        f32x16_t s[2];
        {
            frag8_t kf[2][KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    kf[b][ks] = *reinterpret_cast<const frag8_t*>(kb + b * 32 * KROW + krow + (((2 * ks + hi) ^ kswz) * 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[b][e] = 0.f;
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int b = 0; b < 2; ++b) s[b] = mfma32(kf[b][ks], qf[ks], s[b]);
            __builtin_amdgcn_s_setprio(0);
        }
        // ---- mask + online softmax: this lane's scores are keys kt + 32 b + (r & 3) + 8 (r >> 2) + 4 hi of query q ----
        // (masking only on the causal diagonal and on the ragged last tile: one wave-uniform branch)
        float mx = NEG_BIG;
        if ((kt + 64 > Tk) || (CAUSAL && kt + 63 > sd.pos0 + r0)) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = s[b][r] * sc2;
                    if (key >= Tk || (CAUSAL && key > qpos)) v = NEG_BIG;
                    s[b][r] = v; mx = fmaxf(mx, v);
                }
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[b][r] *= sc2; mx = fmaxf(mx, s[b][r]); }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(mq, mx);
        if (__any(mn > mq)) {                                 // lazily: the running maximum moves on few tiles
            const float corr = __builtin_amdgcn_exp2f(mq - mn);
            lq *= corr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[nt][e] *= corr;
            mq = mn;
        }
        frag8_t pf[4];                                       // step st = 2 b + j: keys 32 b + 16 j ..
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            float e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { e[i] = __builtin_amdgcn_exp2f(s[st >> 1][8 * (st & 1) + i] - mq); lq += e[i]; }    // masked: exp2(-1e30) = 0
            const uint4 pk = make_uint4(cvt_pk(e[0], e[1]), cvt_pk(e[2], e[3]), cvt_pk(e[4], e[5]), cvt_pk(e[6], e[7]));
            pf[st] = __builtin_bit_cast(frag8_t, pk);
        }
        // ---- O^T += V^T P^T: the 2 NT transposed reads of step st + 1 are issued before the NT MFMAs of step st ----
        uint2 va[2 * NT], vn[2 * NT];
        const uint32_t a0 = (uint32_t)(uintptr_t)vb + vrd;
        auto issue = [&](int st, uint2 (&v)[2 * NT]) {       // st is a literal at every call site
            const uint32_t a = a0 + ((st >> 1) * DSUB) * VSUB + (16 * (st & 1)) * 32;
            if constexpr (NT == 4) {
                asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:256\n\t"
                             "ds_read_b64_tr_b16 %2, %8 offset:2304\n\tds_read_b64_tr_b16 %3, %8 offset:2560\n\t"
                             "ds_read_b64_tr_b16 %4, %8 offset:4608\n\tds_read_b64_tr_b16 %5, %8 offset:4864\n\t"
                             "ds_read_b64_tr_b16 %6, %8 offset:6912\n\tds_read_b64_tr_b16 %7, %8 offset:7168"
                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                             : "v"(a) : "memory");
            } else {
                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:256\n\t"
                             "ds_read_b64_tr_b16 %2, %4 offset:2304\n\tds_read_b64_tr_b16 %3, %4 offset:2560"
                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a) : "memory");
            }
        };
        auto landed = [&](uint2 (&v)[2 * NT], auto more) {   // wait-only statement that (re)defines the set: its consumers cannot move above it
            if constexpr (NT == 4) {
                if constexpr (decltype(more)::value) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
            } else {
                if constexpr (decltype(more)::value) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
            }
        };
        auto pv = [&](int st, uint2 (&v)[2 * NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const frag8_t vf = __builtin_bit_cast(frag8_t, make_uint4(v[2 * nt].x, v[2 * nt].y, v[2 * nt + 1].x, v[2 * nt + 1].y));
                o[nt] = mfma32(vf, pf[st], o[nt]);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2);
        issue(0, va);
        issue(1, vn); landed(va, std::true_type{}); pv(0, va); __builtin_amdgcn_sched_barrier(0);
        issue(2, va); landed(vn, std::true_type{}); pv(1, vn); __builtin_amdgcn_sched_barrier(0);
        issue(3, vn); landed(va, std::true_type{}); pv(2, va); __builtin_amdgcn_sched_barrier(0);
        landed(vn, std::false_type{}); pv(3, vn);
        __builtin_amdgcn_s_setprio(0);
    };
    if constexpr (!PACK) {
        stage(0, 0, sd.slot, Tk);
        for (int kt = 0, buf = 0; kt < kend; kt += 64, buf ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // tile `buf` landed for everyone; tile buf ^ 1 no longer read
            if (kt + 64 < kend) stage(kt + 64, buf ^ 1, sd.slot, Tk);
            if (wave_has_rows && !(CAUSAL && kt > wave_last_pos)) compute(kt, buf);
        }
    } else {
        // steps: the shared tiles [0, S) (owner -1: every wave computes), then for each sequence w of the pack its tiles [S, Tk_w)
        int slot4[4], tk4[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const SeqDesc sw = seqs[pk[w] >= 0 ? pk[w] : pk[0]];
            slot4[w] = sw.slot; tk4[w] = pk[w] >= 0 ? sw.pos0 + sw.Tq : 0;
        }
        const int S = (sd.plen / 64) * 64;
        auto first_owner = [&](int from) { int w = from; while (w < 4 && tk4[w] <= S) ++w; return w; };
        int owner = S > 0 ? -1 : first_owner(0), kt = S > 0 ? 0 : S;
        auto issue = [&](int o, int k, int buf) {
            if (o < 0) stage(k, buf, sd.slot, 1 << 30);      // prefix keys only: no clamp, no own pool
            else stage(k, buf, o == 0 ? slot4[0] : (o == 1 ? slot4[1] : (o == 2 ? slot4[2] : slot4[3])), o == 0 ? tk4[0] : (o == 1 ? tk4[1] : (o == 2 ? tk4[2] : tk4[3])));
        };
        if (owner < 4) issue(owner, kt, 0);
        for (int buf = 0; owner < 4; buf ^= 1) {
            int no = owner, nk = kt + 64;                      // the step after this one
            if (owner < 0) { if (nk >= S) { no = first_owner(0); nk = S; } }
            else if (nk >= (owner == 0 ? tk4[0] : (owner == 1 ? tk4[1] : (owner == 2 ? tk4[2] : tk4[3])))) { no = first_owner(owner + 1); nk = S; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (no < 4) issue(no, nk, buf ^ 1);
            if (wave_has_rows && (owner < 0 || owner == wave) && !(kt > wave_last_pos)) compute(kt, buf);
            owner = no; kt = nk;
        }
    }
    // ---- finish: this lane holds O[q][32 nt + (r & 3) + 8 (r >> 2) + 4 hi]; the row sum is split over lanes q and q + 32.
    // Stored from the registers that is 16 stores of 8 B per lane at a row stride (every store instruction touches 32 rows, every
    // 128-byte line is written 8 times): 14 % of the kernel at 611-token prefixes.  So the wave's 32 x D tile goes through LDS
    // (the K / V buffers are free now) and leaves as whole rows, 16 B per lane, 4 rows per instruction.
    __syncthreads();                                          // every wave is done with the tile buffers
    if (!wave_has_rows) return;
    const float inv = 1.f / (lq + __shfl_xor(lq, 32));
    constexpr int OP = D * 2 + 16;                            // row pitch of the staging tile (16-byte aligned, off the 256-byte bank period)
    char* ot = lds + wave * (32 * OP);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const uint2 w = make_uint2(cvt_pk(o[nt][4 * r4] * inv, o[nt][4 * r4 + 1] * inv), cvt_pk(o[nt][4 * r4 + 2] * inv, o[nt][4 * r4 + 3] * inv));
            *reinterpret_cast<uint2*>(ot + ql * OP + (32 * nt + 8 * r4 + 4 * hi) * 2) = w;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // same wave reads it back: LDS program order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int CPRO = D / 8;                               // 16-byte chunks per output row
    constexpr int RPI = 64 / CPRO;                            // rows per store instruction (4 at D = 128, 8 at D = 64)
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + lane / CPRO, c = lane % CPRO;
        const int qr = r0 + row;
        if (qr < sd.Tq) {
            const uint4 v = *reinterpret_cast<const uint4*>(ot + row * OP + c * 16);
            *reinterpret_cast<uint4*>(out + ((size_t)(sd.q_row0 + qr) * H + head) * D + c * 8) = v;
        }
    }
#endif
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
            v = e2f(f2e(v + b));
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
        out[(size_t)patch * Kp + col] = (uint16_t)f2e(v);
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

// ------------------------------------------------------------------ attention probabilities of one prompt (materialised on request)
// The reference's POPE driver asks generate() for the attention maps (output_attentions=True, llava_calibrate.py:175) and reads ONE of
// them: model_outputs['attentions'][0][-1] - step 0, last layer, [1, H, T, T] - to average it (:180-182).  The flash kernels never
// build that matrix; this kernel does, for one sequence, from the rotated q of the last layer and the K already in the caches:
// out[h][i][t] = softmax_t(q_i . k_t * scale) over t <= pos0 + i in fp32, rounded to the model dtype (HF's eager attention:
// fp32 softmax, then `.to(query.dtype)`), zeros behind the diagonal.  One block per (query row, head); a 16-lane group per key.
__global__ void __launch_bounds__(256) attn_probs_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ kpre,
                                                         SeqDesc sd, uint16_t* __restrict__ out, int H, int Hkv, long long slot_stride, int t_max,
                                                         long long pre_stride, int pre_tmax, float scale) {
    constexpr int D = 128;
    extern __shared__ float sc[];                                  // Tk scores of this (row, head)
    __shared__ float red[8];
    const int i = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15;
    const int Tk = sd.pos0 + sd.Tq, n = sd.pos0 + i + 1;           // keys this query sees
    const int kvh = head / (H / Hkv);
    const uint4 qv = *reinterpret_cast<const uint4*>(q + ((size_t)(sd.q_row0 + i) * H + head) * D + j * 8);
    const uint16_t* k_pre = kpre + (size_t)sd.pslot * pre_stride + (size_t)kvh * pre_tmax * D + j * 8;
    const uint16_t* k_own = kc + (size_t)sd.slot * slot_stride + (size_t)kvh * t_max * D + j * 8 - (size_t)sd.plen * D;
    float mx = -INFINITY;
    for (int t0 = wave * 4 + g; t0 < n; t0 += 16) {
        const uint4 kv = *reinterpret_cast<const uint4*>((t0 < sd.plen ? k_pre : k_own) + (size_t)t0 * D);
        float s_ = dot2(qv.x, kv.x, 0.f); s_ = dot2(qv.y, kv.y, s_); s_ = dot2(qv.z, kv.z, s_); s_ = dot2(qv.w, kv.w, s_);
        s_ += __shfl_xor(s_, 1); s_ += __shfl_xor(s_, 2); s_ += __shfl_xor(s_, 4); s_ += __shfl_xor(s_, 8);
        s_ *= scale;
        if (j == 0) sc[t0] = s_;
        mx = fmaxf(mx, s_);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int t = tid; t < n; t += 256) { const float e = expf(sc[t] - mx); sc[t] = e; sum += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
    uint16_t* orow = out + ((size_t)head * sd.Tq + i) * Tk;
    for (int t = tid; t < Tk; t += 256) orow[t] = t < n ? (uint16_t)f2e(sc[t] * inv) : (uint16_t)0;
}

inline int ok() { return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH; }

}  // namespace VDD_ELEM_NS
}  // namespace

using namespace VDD_ELEM_NS;

extern "C" {

VDD_HIDDEN int VDD_IMPL(vdd_flash_attention)(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                        const int32_t* seqs, void* out, int n_seq, int max_tq, int H, int Hkv, int D, int64_t slot_stride, int t_max,
                        int64_t prefix_stride, int prefix_tmax, float scale, int causal, void* stream) {
    if (n_seq <= 0 || max_tq <= 0) return VDD_OK;
    if (!q || !k_cache || !v_cache || !k_prefix || !v_prefix || !seqs || !out || (D != 128 && D != 64) || H % Hkv != 0) return VDD_ERR_INVALID_ARG;
    const int nx = (max_tq + 127) / 128;
    dim3 grid((unsigned)(((long long)H * n_seq + 7) / 8 * 8 * nx)), block(256);
    hipStream_t st = (hipStream_t)stream;
    auto Q = (const uint16_t*)q; auto K = (const uint16_t*)k_cache; auto V = (const uint16_t*)v_cache; auto O = (uint16_t*)out;
    auto S = (const SeqDesc*)seqs; auto KP = (const uint16_t*)k_prefix; auto VP = (const uint16_t*)v_prefix;
    const long long ps = (long long)prefix_stride;
#define VDD_FLASH(DD, CC)                                                                                                        \
    {                                                                                                                            \
        auto kfn = flash_attn2_kernel<DD, CC, false>;                                                                            \
        constexpr int smem = 2 * (64 * DD * 2 + 2 * (DD / 16) * 1152);                                                           \
        static bool attr_set = false;                                                                                            \
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; } \
        hipLaunchKernelGGL(kfn, grid, block, smem, st, Q, K, V, KP, VP, S, O, H, Hkv, (long long)slot_stride, t_max, ps, prefix_tmax, scale, nx, n_seq, (const int4*)nullptr); \
    }
    if (D == 128) { if (causal) VDD_FLASH(128, true) else VDD_FLASH(128, false) }
    else { if (causal) VDD_FLASH(64, true) else VDD_FLASH(64, false) }
#undef VDD_FLASH
    return ok();
}

VDD_HIDDEN int VDD_IMPL(vdd_flash_attention_packed)(const void* q, const void* k_cache, const void* v_cache, const void* k_prefix, const void* v_prefix,
                               const int32_t* seqs, const int32_t* packs, void* out, int n_packs, int H, int Hkv, int D,
                               int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, float scale, void* stream) {
    if (n_packs <= 0) return VDD_OK;
    if (!q || !k_cache || !v_cache || !k_prefix || !v_prefix || !seqs || !packs || !out || D != 128 || H % Hkv != 0) return VDD_ERR_INVALID_ARG;
    auto kfn = flash_attn2_kernel<128, true, true>;
    constexpr int smem = 2 * (64 * 128 * 2 + 2 * (128 / 16) * 1152);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
    dim3 grid((unsigned)(((long long)H * n_packs + 7) / 8 * 8)), block(256);
    hipLaunchKernelGGL(kfn, grid, block, smem, (hipStream_t)stream, (const uint16_t*)q, (const uint16_t*)k_cache, (const uint16_t*)v_cache,
                       (const uint16_t*)k_prefix, (const uint16_t*)v_prefix, (const SeqDesc*)seqs, (uint16_t*)out, H, Hkv, (long long)slot_stride,
                       t_max, (long long)prefix_stride, prefix_tmax, scale, 1, n_packs, (const int4*)packs);
    return ok();
}

VDD_HIDDEN int VDD_IMPL(vdd_attention_probs)(const void* q, const void* k_cache, const void* k_prefix, const int32_t* seq, void* out, int H, int Hkv,
                                             int D, int64_t slot_stride, int t_max, int64_t prefix_stride, int prefix_tmax, float scale, void* stream) {
    if (!q || !k_cache || !k_prefix || !seq || !out || D != 128 || H <= 0 || Hkv <= 0 || H % Hkv != 0) return VDD_ERR_INVALID_ARG;
    SeqDesc sd;                                                   // HOST descriptor of the one sequence: {q_row0, Tq, pos0, slot, prefix_slot, prefix_len}
    sd.q_row0 = seq[0]; sd.Tq = seq[1]; sd.pos0 = seq[2]; sd.slot = seq[3]; sd.pslot = seq[4]; sd.plen = seq[5];
    if (sd.Tq <= 0) return VDD_OK;
    const int Tk = sd.pos0 + sd.Tq;
    if (sd.pos0 < 0 || sd.plen < 0 || sd.plen > Tk || Tk > 16384 || H > 65535) return VDD_ERR_INVALID_ARG;
    // one score row of Tk floats in dynamic LDS (+ 32 B static): up to 64 KiB needs no opt-in, 16,384 keys (64 KiB + 32 B) do
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_probs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * (int)sizeof(float)); attr_set = true; }
    hipLaunchKernelGGL(attn_probs_kernel, dim3(sd.Tq, H), dim3(256), (size_t)Tk * sizeof(float), (hipStream_t)stream, (const uint16_t*)q,
                       (const uint16_t*)k_cache, (const uint16_t*)k_prefix, sd, (uint16_t*)out, H, Hkv, (long long)slot_stride, t_max,
                       (long long)prefix_stride, prefix_tmax, scale);
    return ok();
}

VDD_HIDDEN int VDD_IMPL(vdd_layernorm)(const void* x, const void* w, const void* b, void* y, int M, int d, float eps, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!x || !w || !b || !y || d % 8 != 0 || d > 4096) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(layernorm_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)w,
                       (const uint16_t*)b, (uint16_t*)y, d, eps);
    return ok();
}

VDD_HIDDEN int VDD_IMPL(vdd_bias_act)(const void* x, const void* bias, void* y, int64_t M, int d, int act, void* stream) {
    if (M <= 0) return VDD_OK;
    if (!x || !y || d % 8 != 0 || act < 0 || act > 2) return VDD_ERR_INVALID_ARG;
    long long n8 = (long long)M * (d / 8);
    int blocks = (int)((n8 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bias_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)bias,
                       (uint16_t*)y, (long long)M, d, act);
    return ok();
}

VDD_HIDDEN int VDD_IMPL(vdd_vit_im2col)(const void* images, int dtype, void* patches, int n, int S, int P, int Kp, void* stream) {
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

VDD_HIDDEN int VDD_IMPL(vdd_vit_assemble)(const void* emb, const void* cls, const void* pos, void* out, int n, int T, int width, void* stream) {
    if (n <= 0) return VDD_OK;
    if (!emb || !cls || !pos || !out || width % 8 != 0 || T < 2) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(n * T), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)emb, (const uint16_t*)cls,
                       (const uint16_t*)pos, (uint16_t*)out, T, width);
    return ok();
}

VDD_HIDDEN int VDD_IMPL(vdd_vit_qkv_split)(const void* qkv, void* q, void* k_cache, void* v_cache, int n, int T, int H, int D, int64_t slot_stride, int t_max,
                      int parts, void* stream) {
    if (n <= 0) return VDD_OK;
    if (!qkv || (!q && parts == 3) || !k_cache || !v_cache || D % 8 != 0 || T > t_max || (parts != 2 && parts != 3)) return VDD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(vit_qkv_split_kernel, dim3(n * T), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, (uint16_t*)q,
                       (uint16_t*)k_cache, (uint16_t*)v_cache, T, H, D, (long long)slot_stride, t_max, parts);
    return ok();
}

VDD_HIDDEN int VDD_IMPL(vdd_add)(const void* a, const void* b, void* out, int64_t n, void* stream) {
    if (n <= 0) return VDD_OK;
    if (!a || !b || !out || n % 8 != 0) return VDD_ERR_INVALID_ARG;
    const long long n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)out, n8);
    return ok();
}

}  // extern "C"
