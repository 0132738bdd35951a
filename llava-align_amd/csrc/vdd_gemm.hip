// Row-batched projection GEMM for gfx950:  Y[M, N] = epilogue(X[M, K] . W[N, K]^T), bf16 in / fp32 accumulate / bf16 out.
//
// This is every dense contraction of the path above 8 rows: the four decoder projections + lm_head at the decode batch
// (M = questions x branches, 384-1536 rows), the same projections at prefill size (M = 20k-40k packed prompt tokens), and
// the ViT / projector GEMMs.  It replaces the eager nn.Linear calls under
// experiments/llava/model/language_model/llava_llama.py:88-103 (HF LlamaModel), multimodal_encoder/clip_encoder.py:39-51
// (HF CLIPVisionModel) and multimodal_projector/builder.py:33-46 of the reference.
//
// Design (MI355X: 256 CUs, 160 KiB LDS, 512-entry unified VGPR/AGPR file per SIMD lane, MFMA 32x32x16 bf16 = 32 cycles/SIMD):
//   * macro tile BM x BN (256 x 256 by default), 8 waves as WM x WN; a wave owns a (BM/WM) x (BN/WN) block as 32x32 MFMA
//     tiles with W as the MFMA "A" operand and X as the "B" operand, so the accumulator of a lane is 4 CONSECUTIVE output
//     columns of one output row per register quad (8-byte stores; the SwiGLU pair of a feature sits in the same lane);
//   * both operands are K-contiguous, so both go HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA: no staging
//     VGPRs, no ds_write pass).  One wave instruction moves 8 rows x 128 B: full 128-byte lines on the memory side
//     (fragment-shaped 32-byte pieces cost 4x the address-coalescer cycles), a lane-linear 1-KiB image on the LDS side;
//   * LDS tile = [BM + BN rows][64 k] bf16, row-major 128-B rows, the 16-B chunk index XOR-swizzled with (row >> 1) & 7.
//     LDS-DMA cannot scatter, so the permutation is applied to the SOURCE address of each lane and again on the fragment
//     reads; a ds_read_b128 lane group (16 rows x one chunk) then covers all 16 bank slots: conflict-free;
//   * K-tile = 64, two LDS buffers (2 x 64 KiB at 256 x 256), ONE barrier per K-tile placed behind the last fragment read
//     of the tile (the middle of its MFMAs): fragments are read two 16-deep k-steps ahead of their MFMAs, also across the
//     tile boundary, so MFMAs always issue from registers; the LDS-DMA of tile t+2 goes out right behind that barrier, one
//     instruction per MFMA, and has a whole tile of MFMAs to land;
//   * PERSISTENT stream-K: the grid is one workgroup per CU; the iteration space (tiles x 128-deep K units, tiles in an
//     XCD-contiguous, L2-grouped order) is cut into equal contiguous ranges, so 144 or 258 tiles load 256 CUs evenly (the
//     decode batch is where tile-count quantisation costs more than the inner loop: 768 x 12288 is 144 tiles of 256 x 256).
//     A tile cut across workgroups is finished by the workgroup that owns its FIRST K unit - that part is the LAST thing
//     that workgroup computes, while the other parts are the FIRST thing their workgroups compute, so the fp32 partials
//     (one 256-KiB slab per workgroup, written once) are long complete when the finisher asks for them.  No fences: the
//     slabs are written with write-through (sc1) stores and read with agent-scope (sc1) loads, the per-tile arrival
//     counter is a relaxed agent-scope atomic (an agent-scope release / acquire FENCE writes back / drops the whole L2 of
//     the XCD under the 31 workgroups still streaming through it);
//   * epilogues in registers: bias, bias + quick-GELU / GELU (ViT, projector), bias + residual (ViT), SwiGLU (gate and up
//     rows of one feature are interleaved into the same W tile, so silu(g) * u never round-trips through HBM); the bf16
//     tile then leaves through 4 KiB of LDS per wave (behind the two K-tile buffers: the workgroup owns all 160 KiB) as
//     whole rows, 16 B per lane, with NONTEMPORAL stores - Y must not displace the operand panels the workgroups of an
//     XCD share in its L2 (8-byte quads at a row stride cost 5-12 % of a prefill-size GEMM).
// Bound: MFMA (2.5 PF/s dense bf16).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "vdd_elem.h"

namespace {
namespace VDD_ELEM_NS {
using namespace vdd_elem;

typedef ex8_t frag8_t;                                       // MFMA A / B operand in the element type of this instantiation
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

enum { EPI_NONE = VDD_GEMM_NONE, EPI_BIAS = VDD_GEMM_BIAS, EPI_BIAS_QUICK_GELU = VDD_GEMM_BIAS_QUICK_GELU,
       EPI_BIAS_GELU = VDD_GEMM_BIAS_GELU, EPI_SWIGLU = VDD_GEMM_SWIGLU, EPI_BIAS_RESID = VDD_GEMM_BIAS_RESID,
       EPI_SLABS = 6 };          // internal: schedule 3 of vdd_gemm (split-K slabs) - its own instantiation: the 256 x 256 tile sits at the
                                 // register cliff (24 - 38 spilled VGPRs), and a run-time slab branch in every instance cost it 200 more

struct GemmArgs {
    const uint16_t* X; const uint16_t* W; uint16_t* Y;
    const uint16_t* bias; const uint16_t* resid;
    float* partial;              // [P][BM * BN] fp32: the partial tile a workgroup hands to a finisher
    int* counter;                // [Mt * Nt] arrival counters, zero between launches (the finisher resets its tile's)
    int M, N, K;                 // N: OUTPUT columns (SwiGLU: the F features; W then has 2 F rows, gate rows first)
    long long ldx, ldw, ldy, ldr;
    int Mt, Nt, GM;              // tile counts, row-tiles per L2 group
    int UP, U, P;                // K units (128 deep) per tile, units in total, workgroups
    int dp_rounds;               // whole-tile rounds before the stream-K part (host-chosen schedule)
    int stage_out;               // Y rows are 16-byte aligned: the epilogue may store whole rows out of LDS
    float* slabs;                // split-K slab mode (slab_S > 0): fp32 partial products [slab_S][M][N], one (tile, K part) per workgroup, no
    int slab_S;                  // fix-up - the consumer (vdd_rmsnorm's delta_slabs) adds the slabs
};

// K-tile buffers of a tile shape.  Two for the compute-bound shapes (tile t + 1 lands under the MFMAs of tile t).  The shapes the
// tuner picks for a few dozen to ~250 rows of activations are W STREAMS - one or two row-tiles, every CU pulling its own column
// panel of W from HBM - and what bounds those is the bytes a CU has in flight: with two buffers ONE K-tile (16 KiB of W for the
// 192 x 128 shape), i.e. one HBM round trip (2 us under load) per K-tile whose MFMAs take 0.35 us.  They get as many buffers as
// the 160 KiB of LDS hold beside the epilogue's staging blocks.
constexpr int LDS_BYTES = 160 * 1024;
constexpr int gemm_staging_bytes(int BN, int WM, int WN) { return ((BN / WN) % 64 == 0) ? WM * WN * 4096 : 0; }
constexpr int gemm_stages(int BM, int BN, int WM, int WN) {
    // 64 x 128: up to 64 rows of activations without padding the row tile to 128 - 24 KiB per K-tile buffer, six of them: four K-tiles (64 KiB of W
    // per CU) in flight behind the two being worked on
    // (32 x 128, up to 32 rows: 20 KiB per buffer, eight of them = 96 KiB of W in flight; never more buffers than the 6-bit vmcnt can count)
    if (BM <= 64 && BN == 128) {
        const int n = (LDS_BYTES - gemm_staging_bytes(BN, WM, WN)) / ((BM + BN) * 128), cap = 63 / ((BM + BN) / 8 / (WM * WN)) + 1;
        const int m = n > 8 ? 8 : n;
        return m > cap ? cap : m;
    }
    if (BM == 64) return 3;
    if (BN == 128 && BM <= 192) { const int n = (LDS_BYTES - gemm_staging_bytes(BN, WM, WN)) / ((BM + BN) * 128); return n > 5 ? 5 : n; }
    return 2;
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ void __launch_bounds__(WM * WN * 64) gemm_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub: it drops a kernel whose body holds LDS-DMA builtins
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    constexpr int XIMG = BM / 8, WIMG = BN / 8;            // 1-KiB LDS-DMA images (8 rows x 128 B) per K-tile
    constexpr int XJ = XIMG / NW, WJ = WIMG / NW;          // images per wave
    constexpr int BUF = (BM + BN) * 128;                   // bytes of one K-tile buffer
    constexpr int NSTG = gemm_stages(BM, BN, WM, WN);      // K-tile buffers (> 2: a W stream, see the K pipeline below)
    constexpr int NMMA = NI * MI, NRD = NI + MI, NLD = XJ + WJ;
    // (the 192 x 256 tile of the decode batch only: down-projection 134 -> 120 us, o 57 -> 53 on cold weights; the 256 x 256 tile lost 5 - 12 % with it
    //  at 1,536 rows - 13 more spilled registers at the cliff - and gained nothing at prefill size, where the panels are L2 hits)
    constexpr bool WAHEAD = NSTG == 2 && gemm_staging_bytes(BN, WM, WN) >= BN * 128 && BM == 192 && BN == 256;
    static_assert(XIMG % NW == 0 && WIMG % NW == 0 && TM % 32 == 0 && TN % 32 == 0, "tile / wave shape");
    static_assert(EPI != EPI_SWIGLU || NI % 2 == 0, "SwiGLU pairs gate/up 32-column blocks inside a wave");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;

    // fragment addresses: row * 128 + ((kk * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 16
    const int c0 = ((lane >> 5) ^ ((lane >> 1) & 7)) * 16;
    const int xrow = (wr * TM + (lane & 31)) * 128 + c0;
    const int wrow = (BM + wc * TN + (lane & 31)) * 128 + c0;

    // ---- this workgroup's work list.  Whole tiles first: round i of the data-parallel part hands tile i * P + rng to
    // workgroup rng, and workgroup b runs on XCD b % 8, so rng is XCD-contiguous: the 32 CUs of an XCD work on 32 neighbouring
    // tiles at any time (shared X row-panels / W column-panels in its L2).  Then the stream-K part: the tiles that do not
    // fill a round, as one contiguous range of 128-deep K units per workgroup.
    // Split-K slab mode: the work items are (tile, K part) pairs, handed out like whole tiles; nothing is left for stream-K.
    constexpr bool SLAB = EPI == EPI_SLABS;
    const int S_ = SLAB ? a.slab_S : 1, T = a.Mt * a.Nt, items = SLAB ? T * S_ : T;
    const int rounds = SLAB ? (items + a.P - 1) / a.P : a.dp_rounds, dp_tiles = SLAB ? T : min(T, rounds * a.P);
    const long long SU = (long long)(T - dp_tiles) * a.UP;            // stream-K units
    const int Psk = (int)min((long long)a.P, SU);                      // never more ranges than units: no empty range
    // n work items for the P workgroups of a round: XCD x (= blockIdx % 8) takes a contiguous run of ceil / floor (n / 8) items,
    // so a partial round still spreads over all eight L2s and neighbours stay together; -1 = nothing for this workgroup
    const int bx = blockIdx.x & 7, by = blockIdx.x >> 3;
    auto spread = [&](int n) {
        const int q = n >> 3, r = n & 7, cnt = q + (bx < r ? 1 : 0);
        return by < cnt ? (bx < r ? bx * (q + 1) : r * (q + 1) + (bx - r) * q) + by : -1;
    };
    const int rng = spread(Psk);                                       // stream-K range of this workgroup
    long long u = 0, u_end = 0;
    if (rng >= 0) { u = (long long)rng * SU / Psk; u_end = (long long)(rng + 1) * SU / Psk; }
    auto owner = [&](long long uu) { return (int)(((uu + 1) * Psk + SU - 1) / SU - 1); };     // stream-K range that holds unit uu
    int dp_i = 0;
    int L = 0, ku0 = 0, ku1 = 0, part = 0;
    auto advance = [&]() -> bool {
        while (dp_i < rounds) {
            const int j = spread(min(a.P, items - dp_i * a.P));
            ++dp_i;
            if (j >= 0) {
                const int it = (dp_i - 1) * a.P + j;
                if constexpr (SLAB) { L = it / S_; part = it - L * S_; ku0 = (int)((long long)part * a.UP / S_); ku1 = (int)((long long)(part + 1) * a.UP / S_); }
                else { L = it; ku0 = 0; ku1 = a.UP; }
                return true;
            }
        }
        if (u < u_end) {
            const int ls = (int)(u / a.UP);
            L = dp_tiles + ls; ku0 = (int)(u - (long long)ls * a.UP);
            ku1 = (int)min((long long)a.UP, ku0 + (u_end - u));
            u += ku1 - ku0;
            return true;
        }
        return false;
    };
    // per-segment state: tile origin, LDS-DMA descriptors (one buffer descriptor per operand, per-lane byte offsets constant over K)
    int tn = 0, m0 = 0, n0 = 0, nk = 0;
    __amdgpu_buffer_rsrc_t rx, rw;
    uint32_t xoff[XJ], woff[WJ];
    auto setup = [&]() {
        // tile order: groups of GM row-tiles x all column tiles, row-tile fastest (neighbours share a W column panel)
        const int per_group = a.GM * a.Nt, gid = L / per_group, first_m = gid * a.GM;
        const int gsz = min(a.Mt - first_m, a.GM), in_g = L - gid * per_group;
        const int tm = first_m + in_g % gsz;
        tn = in_g / gsz;
        m0 = tm * BM; n0 = tn * BN;
        const int kbeg = ku0 * 128;
        nk = (ku1 - ku0) * 2;
        const int rows_x = min(BM, a.M - m0);
        const uint16_t* xb = a.X + (size_t)m0 * a.ldx + kbeg;
        const uint16_t* wb;
        int rows_w;
        if constexpr (EPI == EPI_SWIGLU) { wb = a.W + (size_t)(tn * (BN / 2)) * a.ldw + kbeg; rows_w = BN; }
        else { wb = a.W + (size_t)n0 * a.ldw + kbeg; rows_w = min(BN, a.N - n0); }
        rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7fffffff, 0x00020000);
        rw = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int row = (j * NW + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            xoff[j] = (uint32_t)min(row, rows_x - 1) * (uint32_t)(a.ldx * 2) + c * 16;
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int row = (j * NW + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int src;
            if constexpr (EPI == EPI_SWIGLU) {      // tile rows: [g 0-31 | u 0-31 | g 32-63 | u 32-63 | ...] of features tn*BN/2 ..
                const int blk = row >> 5;
                src = (blk & 1) * a.N + (blk >> 1) * 32 + (row & 31);
            } else {
                src = min(row, rows_w - 1);
            }
            woff[j] = (uint32_t)src * (uint32_t)(a.ldw * 2) + c * 16;
        }
    };
    auto stage = [&](int t, int buf) {
        const int so = t * 128;
#pragma unroll
        for (int j = 0; j < XJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(lds + buf * BUF + (j * NW + wave) * 1024), 16, xoff[j], so, 0, 0);
#pragma unroll
        for (int j = 0; j < WJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(lds + buf * BUF + (XIMG + j * NW + wave) * 1024), 16, woff[j], so, 0, 0);
    };

    auto stage_x = [&](int t, int buf) {
        const int so = t * 128;
#pragma unroll
        for (int j = 0; j < XJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(lds + buf * BUF + (j * NW + wave) * 1024), 16, xoff[j], so, 0, 0);
    };
    auto stage_w = [&](int t, int buf) {
        const int so = t * 128;
#pragma unroll
        for (int j = 0; j < WJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(lds + buf * BUF + (XIMG + j * NW + wave) * 1024), 16, woff[j], so, 0, 0);
    };
    bool more = advance();
    auto stage_first = [&]() {
        stage(0, 0); stage(1, 1);
        if constexpr (NSTG >= 3) {
#pragma unroll
            for (int i = 2; i < NSTG; ++i) if (nk > i) stage(i, i);
        }
    };
    if (more) { setup(); stage_first(); }
    while (more) {
        const int cL = L, cku0 = ku0, cku1 = ku1, ctn = tn, cm0 = m0, cn0 = n0, cpart = part;      // this segment (the state moves on to the next one below)
        f32x16_t acc[NI][MI];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // ---- K pipeline.  Fragments are read TWO 16-deep k-steps ahead of their MFMAs (sets kk = 0..3, three live at a
        // time), one LDS read (and, behind the barrier, one LDS-DMA of tile t+2) issued behind each MFMA.
        // (the first two K-tiles of this segment were staged before the previous segment's epilogue)
        if constexpr (NSTG == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        auto interleave = [&](bool with_dma, bool with_reads) {             // MFMA, [DMA], [read], MFMA, ...
#pragma unroll
            for (int i = 0; i < NMMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (with_dma && i < NLD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (with_reads && i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        auto interleave_n = [&](int n_dma, bool with_reads) {               // MFMA, [DMA] (n_dma of them, one behind each MFMA), [read], MFMA, ...
#pragma unroll
            for (int i = 0; i < NMMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < n_dma) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (with_reads && i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        frag8_t xg[4][MI], wg[4][NI];
        auto rd = [&](int buf, int kk) {
#pragma unroll
            for (int i = 0; i < NI; ++i) wg[kk][i] = *reinterpret_cast<const frag8_t*>(lds + buf * BUF + ((wrow + i * 4096) ^ (kk << 5)));
#pragma unroll
            for (int i = 0; i < MI; ++i) xg[kk][i] = *reinterpret_cast<const frag8_t*>(lds + buf * BUF + ((xrow + i * 4096) ^ (kk << 5)));
        };
        // The wave tile is walked boustrophedon with the X fragment outer and the W fragment inner: consecutive MFMAs differ in ONE operand
        // and the two operand ports take turns (the MFMA pipe's sustained rate on real data depends on how much its inputs toggle:
        // tools/probes/mfma_power_probe.hip - a new operand on one port at every instruction 1.58 PF/s, ports taking turns 1.83).  Worth
        // +0.7 ... 1.9 % over the row-major walk at prefill size (profiles/r04_gemm_mfma_order_ab.jsonl); same accumulation order per accumulator.
        auto mm = [&](int kk) {
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) { const int i = (j & 1) ? NI - 1 - ii : ii; acc[i][j] = mfma32(wg[kk][i], xg[kk][j], acc[i][j]); }
        };
        if constexpr (NSTG >= 3) {
            // A W stream (the 64-row tile: a few dozen rows of activations; the 128-column tiles of up to 192 rows): a handful of MFMAs per
            // wave and k-step, and what bounds the launch is first the bytes in flight - with two buffers one K-tile (40 KiB) per CU,
            // 2.6 TB/s - and then the wave's own instruction stream: fragment reads (0.2 us per tile), LDS-DMA issue (0.3), MFMAs (0.2) run
            // one after the other cost 1 us per 16 KiB of W.  So: NSTG buffers, NSTG - 2 tiles in flight behind the two being worked
            // on, and the pipeline of the compute-bound shapes below (reads two k-steps ahead, ONE barrier per tile, the LDS-DMA of
            // tile t + NSTG into the buffer that barrier frees), with run-time buffer indices.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the previous segment's epilogue stores are counted by vmcnt too)
            __builtin_amdgcn_s_barrier();
            rd(0, 0); rd(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            auto tile = [&](int t, int buf, int nbuf, auto do_stage, auto do_next, auto wait_n) {
                constexpr bool ST = decltype(do_stage)::value, NX = decltype(do_next)::value;
                constexpr int WN_ = decltype(wait_n)::value;
                rd(buf, 2); mm(0); interleave(false, true); __builtin_amdgcn_sched_barrier(0);
                rd(buf, 3); mm(1); interleave(false, true); __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(WN_) : "memory");   // own fragment reads of this tile done; tile t+1 landed
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ST) stage_x(t + NSTG, buf);
                if constexpr (NX) rd(nbuf, 0);
                mm(2); interleave_n(ST ? XJ : 0, NX); __builtin_amdgcn_sched_barrier(0);
                if constexpr (ST) stage_w(t + NSTG, buf);
                if constexpr (NX) rd(nbuf, 1);
                mm(3); interleave_n(ST ? WJ : 0, NX); __builtin_amdgcn_sched_barrier(0);
            };
            using T_ = std::true_type; using F_ = std::false_type;
            using WS_ = std::integral_constant<int, (NSTG - 2) * NLD>; using W0_ = std::integral_constant<int, 0>;
            int t = 0, buf = 0;
            auto nxt = [&](int b) { return b + 1 == NSTG ? 0 : b + 1; };
            for (; t + NSTG < nk; ++t) { tile(t, buf, nxt(buf), T_{}, T_{}, WS_{}); buf = nxt(buf); }
            for (; t + 1 < nk; ++t) { tile(t, buf, nxt(buf), F_{}, T_{}, W0_{}); buf = nxt(buf); }     // the last tiles: everything staged is waited for
            tile(t, buf, buf, F_{}, F_{}, W0_{});
        } else if constexpr (WAHEAD) {
            // Two K-tile buffers + a THIRD W image in the epilogue's staging blocks (a W image of a 256-column tile is exactly their 32 KiB):
            // W runs two tiles ahead of the MFMAs, X one.  At the decode batch W is what comes from HBM (13 GB of weights per step, every
            // K-tile a miss that the eight workgroups sharing the column panel wait for together) while X sits in the L2: with one tile
            // (0.7 us) of lead against ~2 us of latency the cold launch ran 5 - 12 % behind the same launch on resident weights
            // (profiles/r04_gemm_decode_cold_vs_hot.jsonl).  X images alternate between the two buffers (literal indices, as below), W images
            // rotate through three slots (run-time base); the staging blocks are only borrowed between the top-of-segment barrier and the
            // last tile: nothing is in flight into them while an epilogue uses them.
            auto wbase = [&](int sl) { return sl == 2 ? 2 * BUF - BM * 128 : sl * BUF; };            // slot 2: the W image starts at the staging blocks
            auto stage_w3 = [&](int t, int sl) {
                const int so = t * 128;
#pragma unroll
                for (int j = 0; j < WJ; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(lds + wbase(sl) + (XIMG + j * NW + wave) * 1024), 16, woff[j], so, 0, 0);
            };
            auto rd3 = [&](int buf, int sl, int kk) {
#pragma unroll
                for (int i = 0; i < NI; ++i) wg[kk][i] = *reinterpret_cast<const frag8_t*>(lds + wbase(sl) + ((wrow + i * 4096) ^ (kk << 5)));
#pragma unroll
                for (int i = 0; i < MI; ++i) xg[kk][i] = *reinterpret_cast<const frag8_t*>(lds + buf * BUF + ((xrow + i * 4096) ^ (kk << 5)));
            };
            if (nk > 2) stage_w3(2, 2);                           // (behind the barrier above: every wave is through its epilogue)
            rd3(0, 0, 0); rd3(0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            auto tile = [&](int t, int buf, int sl, int sl1, auto do_x, auto do_w, auto do_next, auto wait_n) {      // buf is a literal at every call site
                constexpr bool STX = decltype(do_x)::value, STW = decltype(do_w)::value, NX = decltype(do_next)::value;
                constexpr int WN_ = decltype(wait_n)::value;
                rd3(buf, sl, 2); mm(0); interleave(false, true); __builtin_amdgcn_sched_barrier(0);
                rd3(buf, sl, 3); mm(1); interleave(false, true); __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(WN_) : "memory");   // own reads of tile t done; X and W of tile t+1 landed (W of t+2 may be in flight)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STX) stage_x(t + 2, buf);
                if constexpr (NX) rd3(buf ^ 1, sl1, 0);
                mm(2); interleave_n(STX ? XJ : 0, NX); __builtin_amdgcn_sched_barrier(0);
                if constexpr (STW) stage_w3(t + 3, sl);
                if constexpr (NX) rd3(buf ^ 1, sl1, 1);
                mm(3); interleave_n(STW ? WJ : 0, NX); __builtin_amdgcn_sched_barrier(0);
            };
            using T_ = std::true_type; using F_ = std::false_type;
            using WW_ = std::integral_constant<int, WJ>; using W0_ = std::integral_constant<int, 0>;
            auto nxt = [&](int sl) { return sl == 2 ? 0 : sl + 1; };
            int t = 0, sl = 0;
            for (; t + 4 < nk; t += 2) {
                tile(t, 0, sl, nxt(sl), T_{}, T_{}, T_{}, WW_{}); sl = nxt(sl);
                tile(t + 1, 1, sl, nxt(sl), T_{}, T_{}, T_{}, WW_{}); sl = nxt(sl);
            }
            if (nk - t == 4) {
                tile(t, 0, sl, nxt(sl), T_{}, T_{}, T_{}, WW_{}); sl = nxt(sl);
                tile(t + 1, 1, sl, nxt(sl), T_{}, F_{}, T_{}, WW_{}); sl = nxt(sl);
                t += 2;
            }
            tile(t, 0, sl, nxt(sl), F_{}, F_{}, T_{}, W0_{}); sl = nxt(sl);
            tile(t + 1, 1, sl, sl, F_{}, F_{}, F_{}, W0_{});
        } else {
        rd(0, 0); rd(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        auto tile = [&](int t, int buf, auto do_stage, auto do_next) {           // buf is a literal at every call site
            constexpr bool ST = decltype(do_stage)::value, NX = decltype(do_next)::value;
            rd(buf, 2); mm(0); interleave(false, true); __builtin_amdgcn_sched_barrier(0);
            rd(buf, 3); mm(1); interleave(false, true); __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own fragment reads of this tile done; tile t+1 landed
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // the LDS-DMA of tile t + 2 spread over BOTH post-barrier MFMA groups (X images behind the MFMAs of k-step 2, W images behind
            // those of k-step 3) instead of all of it behind k-step 2: +1 ... 3 % at prefill and decode sizes, A/B in one process
            // (profiles/r04_gemm_ablation.jsonl); W first or X first: the same
            if constexpr (ST) stage_x(t + 2, buf);
            if constexpr (NX) rd(buf ^ 1, 0);
            mm(2); interleave_n(ST ? XJ : 0, NX); __builtin_amdgcn_sched_barrier(0);
            if constexpr (ST) stage_w(t + 2, buf);
            if constexpr (NX) rd(buf ^ 1, 1);
            mm(3); interleave_n(ST ? WJ : 0, NX); __builtin_amdgcn_sched_barrier(0);
        };
        using T_ = std::true_type; using F_ = std::false_type;
        int t = 0;                                  // nk is even: tiles alternate between the two buffers
        for (; t + 2 < nk; t += 2) { tile(t, 0, T_{}, T_{}); tile(t + 1, 1, T_{}, T_{}); }
        tile(t, 0, F_{}, T_{});
        tile(t + 1, 1, F_{}, F_{});
        }
        // nobody reads LDS behind the last barrier of a segment: stage the next segment's first two K-tiles now, so that
        // they land under this segment's epilogue
        more = advance();
        if (more) { setup(); stage_first(); }

        // ---- split-K slab mode: the partial product of this (tile, K part) goes to its slab as it is; the consumer adds the slabs
        if constexpr (SLAB) {
            float* sl = a.slabs + (size_t)cpart * a.M * a.N;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    const int m = cm0 + wr * TM + j * 32 + (lane & 31);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = cn0 + wc * TN + i * 32 + q * 8 + 4 * (lane >> 5);
                        if (m < a.M && n < a.N)
                            *reinterpret_cast<f32x4_t*>(sl + (size_t)m * a.N + n) = f32x4_t{acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                    }
                }
            continue;
        }
        // ---- a tile cut across workgroups
        const bool whole = (cku0 == 0 && cku1 == a.UP);
        if (!whole) {
            if (cku0 != 0) {            // not the first part: hand the partial sums to the finisher
                // write-through (sc1) stores: the slab leaves this XCD's L2 at once, so no release fence (an agent-scope release
                // is a write-back of the whole L2 - with 32 workgroups per XCD publishing 256 KiB each that costs tens of us)
                const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)rng * (BM * BN)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4_t vf = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, vf), rp, (((i * MI + j) * 4 + q) * NT + tid) * 16, 0, 16);
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(a.counter + cL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            // first part (the last thing this workgroup computes): wait for the others, add their partial sums
            const long long tu = (long long)(cL - dp_tiles) * a.UP;
            const int r_last = owner(tu + a.UP - 1), others = r_last - rng;
            if (tid == 0) {
                while (__hip_atomic_load(a.counter + cL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < others) __builtin_amdgcn_s_sleep(8);
                __hip_atomic_store(a.counter + cL, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
            }
            __syncthreads();
            // the slabs are read with agent-scope (sc1) loads, straight from where the write-through stores put them: an acquire
            // FENCE here is a buffer_inv sc1, which drops this XCD's whole L2 under the 31 workgroups still streaming through it
            // The small tiles of the W-stream shapes are cut 4 - 8 ways (32 tiles of the attention-output projection over 256 workgroups),
            // and one slab after the other is one memory round trip (~2 us) each: the finisher of such a tile spent 10 - 15 us here, most
            // of the launch.  So the loads of up to RU slabs go out together (32 - 64 registers each); they are ADDED in slab order as
            // before - the same bits.  (RU = 1 for the large tiles: a slab is 96 - 128 registers there, and a tile is cut in two.)
            constexpr int PER = NI * MI * 16, RU = PER <= 32 ? 4 : PER <= 64 ? 3 : 1;
            auto load_slab = [&](int r, f32x4_t (&v)[NI * MI * 4]) {
                const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)r * (BM * BN)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
                for (int x = 0; x < NI * MI * 4; ++x)
                    v[x] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, (x * NT + tid) * 16, 0, 16));
            };
            auto add_slab = [&](const f32x4_t (&v)[NI * MI * 4]) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4_t& w = v[(i * MI + j) * 4 + q];
                            acc[i][j][q * 4] += w[0]; acc[i][j][q * 4 + 1] += w[1]; acc[i][j][q * 4 + 2] += w[2]; acc[i][j][q * 4 + 3] += w[3];
                        }
            };
            if constexpr (RU > 1) {
                int r = rng + 1;
                for (; r + RU - 1 <= r_last; r += RU) {
                    f32x4_t v[RU][NI * MI * 4];
#pragma unroll
                    for (int g = 0; g < RU; ++g) load_slab(r + g, v[g]);
#pragma unroll
                    for (int g = 0; g < RU; ++g) add_slab(v[g]);
                }
                for (; r <= r_last; ++r) {
                    f32x4_t v[NI * MI * 4];
                    load_slab(r, v);
                    add_slab(v);
                }
            } else {
                for (int r = rng + 1; r <= r_last; ++r) {
                    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(a.partial + (size_t)r * (BM * BN)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < MI; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, (((i * MI + j) * 4 + q) * NT + tid) * 16, 0, 16));
                                acc[i][j][q * 4] += v[0]; acc[i][j][q * 4 + 1] += v[1]; acc[i][j][q * 4 + 2] += v[2]; acc[i][j][q * 4 + 3] += v[3];
                            }
                }
            }
        }

        // ---- epilogue.  acc[i][j][e]: n = wc*TN + i*32 + (e&3) + 8*(e>>2) + 4*(lane>>5),  m = wr*TM + j*32 + (lane&31)
        const int mrow = cm0 + wr * TM + (lane & 31);
        if constexpr (EPI == EPI_SWIGLU) {
            const int f0 = ctn * (BN / 2) + wc * (TN / 2) + 4 * (lane >> 5);
            auto gated = [&](int i, int j, int q, float (&o)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = rnd(acc[i][j][q * 4 + e]), up = rnd(acc[i + 1][j][q * 4 + e]);
                    o[e] = rnd(g / (1.f + __expf(-g))) * up;
                }
            };
            if ((TN == 64) && a.stage_out) {       // 32 feature columns per wave: 64-byte rows through LDS, 16 rows per store instruction
                char* st = lds + NSTG * BUF + wave * 4096;
                const int r = lane & 31, hi = lane >> 5;
#pragma unroll
                for (int j = 0; j < MI; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float o[4];
                        gated(0, j, q, o);
                        *reinterpret_cast<uint2*>(st + r * 64 + ((q ^ ((r >> 2) & 3)) * 16) + hi * 8) = pack4(o[0], o[1], o[2], o[3]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = t * 16 + (lane >> 2), c = lane & 3;
                        const int mm = cm0 + wr * TM + j * 32 + row;
                        const uint4 val = *reinterpret_cast<const uint4*>(st + row * 64 + ((c ^ ((row >> 2) & 3)) * 16));
                        if (mm < a.M) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, val), reinterpret_cast<u32x4_t*>(a.Y + (size_t)mm * a.ldy + ctn * (BN / 2) + wc * (TN / 2) + c * 8));
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NI; i += 2)
#pragma unroll
                    for (int j = 0; j < MI; ++j) {
                        const int m = mrow + j * 32;
                        if (m >= a.M) continue;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float o[4];
                            gated(i, j, q, o);
                            *reinterpret_cast<uint2*>(a.Y + (size_t)m * a.ldy + f0 + (i / 2) * 32 + q * 8) = pack4(o[0], o[1], o[2], o[3]);
                        }
                    }
            }
        } else {
            // one quad of the tile: the 4 consecutive columns n .. n + 3 of row m this lane holds in acc[i][j][4 q ..], epilogue applied
            auto quad = [&](int i, int j, int q, int m, int n, float (&v)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
                if constexpr (EPI != EPI_NONE) {
                    const uint2 bb = *reinterpret_cast<const uint2*>(a.bias + n);
                    v[0] += e2f(bb.x & 0xffffu); v[1] += e2f(bb.x >> 16); v[2] += e2f(bb.y & 0xffffu); v[3] += e2f(bb.y >> 16);
                }
                if constexpr (EPI == EPI_BIAS_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_quick_gelu(rnd(v[e]));
                } else if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_gelu(rnd(v[e]));
                } else if constexpr (EPI == EPI_BIAS_RESID) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(a.resid + (size_t)m * a.ldr + n);
                    v[0] = rnd(v[0]) + e2f(rr.x & 0xffffu); v[1] = rnd(v[1]) + e2f(rr.x >> 16);
                    v[2] = rnd(v[2]) + e2f(rr.y & 0xffffu); v[3] = rnd(v[3]) + e2f(rr.y >> 16);
                }
            };
            // Stored from the accumulator layout a quad is 8 bytes and a store instruction touches 32 rows: every 128-byte line of Y
            // is written by 8 instructions (5-12 % of a prefill-size GEMM, measured with the stores switched off).  With 64-column
            // wave tiles a 32-row block of the wave goes through 4 KiB of LDS behind the two K-tile buffers instead and leaves as
            // whole 128-byte rows, 16 B per lane, 8 rows per instruction (chunks XOR-swizzled with the row: conflict-free reads).
            const bool staged = (TN % 64 == 0) && a.stage_out;
            if (staged) {
                char* st = lds + NSTG * BUF + wave * 4096;
                const int r = lane & 31, hi = lane >> 5;
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    const int m = mrow + j * 32;
#pragma unroll
                    for (int h = 0; h < TN / 64; ++h) {                  // 64-column halves of a 128-column wave tile, one after the other
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int i = h * 2 + ii;
                                const int n = cn0 + wc * TN + i * 32 + q * 8 + 4 * hi;
                                float v[4] = {0.f, 0.f, 0.f, 0.f};
                                if (m < a.M && n < a.N) quad(i, j, q, m, n, v);
                                *reinterpret_cast<uint2*>(st + r * 128 + (((ii * 4 + q) ^ (r & 7)) * 16) + hi * 8) = pack4(v[0], v[1], v[2], v[3]);
                            }
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int row = t * 8 + (lane >> 3), c = lane & 7;
                            const int mm = cm0 + wr * TM + j * 32 + row, nn = cn0 + wc * TN + h * 64 + c * 8;
                            const uint4 val = *reinterpret_cast<const uint4*>(st + row * 128 + ((c ^ (row & 7)) * 16));
                            if (mm < a.M) {
                                uint16_t* yp = a.Y + (size_t)mm * a.ldy + nn;
                                if (nn + 8 <= a.N) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, val), reinterpret_cast<u32x4_t*>(yp));
                                else if (nn + 4 <= a.N) *reinterpret_cast<uint2*>(yp) = make_uint2(val.x, val.y);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) {
                        const int m = mrow + j * 32;
                        if (m >= a.M) continue;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int n = cn0 + wc * TN + i * 32 + q * 8 + 4 * (lane >> 5);
                            if (n >= a.N) continue;
                            float v[4];
                            quad(i, j, q, m, n, v);
                            *reinterpret_cast<uint2*>(a.Y + (size_t)m * a.ldy + n) = pack4(v[0], v[1], v[2], v[3]);       // (nontemporal 8-byte quads: -25 % at prefill size)
                        }
                    }
            }
        }
    }
#endif
}

int g_num_cu = 0;
constexpr size_t COUNTER_BYTES = (size_t)4 << 20;      // arrival counters: one int32 per output tile of a launch

int num_workgroups() {
    if (g_num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_num_cu = n & ~7;       // one workgroup per CU, a multiple of the 8 XCDs
        if (g_num_cu < 8) g_num_cu = 8;
#ifdef VDD_PROBE_BUILD
        // tools/overlap_cu_probe.py: a persistent grid of P < #CUs workgroups on a CU-masked stream (the rest of the chip runs attention)
        if (const char* e = getenv("VDD_PROBE_GEMM_WORKGROUPS")) { const int v = atoi(e); if (v >= 8 && v <= g_num_cu) g_num_cu = v & ~7; }
#endif
    }
    return g_num_cu;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& a0, int epi, int sched, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    GemmArgs a = a0;
    const int ncols = epi == EPI_SWIGLU ? 2 * a.N : a.N;
    a.Mt = (a.M + BM - 1) / BM;
    a.Nt = (ncols + BN - 1) / BN;
    // row-tiles per group of the tile order (a group = GM row-tiles x all column tiles, row-tile fastest: an XCD's 32 workgroups hold GM x 32 / GM
    // tiles).  A/B builds at 39,140 rows (profiles/r04_gemm_tile_order_ab.jsonl): 8 is best for 48 and 16 column tiles (qkv, o, down: 4 costs qkv 1.3 %,
    // 16 costs down 2.6 %), 4 for the 86 column tiles of gate/up (+2.9 %; 16: -5.5 %).  The decode batch (8 row-tiles) keeps one group.
    a.GM = a.Mt < 8 ? a.Mt : ((a.Mt >= 32 && a.Nt >= 80) ? 4 : 8);
    a.UP = a.K / 128;
    const long long U = (long long)a.Mt * a.Nt * a.UP;
    if (U > 0x7fffffffLL) return VDD_ERR_INVALID_ARG;
    a.U = (int)U;
    const int P = num_workgroups();
    a.P = P;
    {   // schedule: whole-tile rounds, then stream-K over what is left.  The stream-K part always spans at least one tile per
        // workgroup when it exists beside full rounds ("two-tile" stream-K), so a tile is cut into 2 (rarely 3) parts.
        const int T = a.Mt * a.Nt, full = T / P, rem = T % P;
        if (a.slab_S > 0) { if (a.slab_S > a.UP) return VDD_ERR_INVALID_ARG; a.dp_rounds = 0; }
        else if (sched == 1) a.dp_rounds = (T + P - 1) / P;       // data-parallel only
        else if (sched == 2) a.dp_rounds = 0;                     // stream-K only
        else a.dp_rounds = rem == 0 ? full : (full > 0 ? full - 1 : 0);
    }
    // Workspace layout: [COUNTER_BYTES of arrival counters | one fp32 partial tile per workgroup].  The counter region has a FIXED
    // size: with the partials placed right behind the Mt x Nt counters of the CURRENT launch, the slabs of a launch with few tiles
    // lay where a later launch with many tiles reads its counters (float bit patterns as arrival counts: a finisher that spins
    // forever on a negative one, or adds partial sums that were never written on a large one).
    if ((size_t)a.Mt * a.Nt * sizeof(int) > COUNTER_BYTES) return VDD_ERR_INVALID_ARG;
    const size_t need = COUNTER_BYTES + (size_t)P * BM * BN * sizeof(float);
    if (!workspace || (size_t)workspace_bytes < need) return VDD_ERR_INVALID_ARG;
    a.counter = (int*)workspace;
    a.partial = (float*)((char*)workspace + COUNTER_BYTES);
    const dim3 grid(P), block(WM * WN * 64);
    const size_t smem = (size_t)gemm_stages(BM, BN, WM, WN) * (BM + BN) * 128 + gemm_staging_bytes(BN, WM, WN);      // + the epilogue's staging blocks
    static_assert(gemm_stages(BM, BN, WM, WN) * (BM + BN) * 128 + gemm_staging_bytes(BN, WM, WN) <= LDS_BYTES, "LDS");
    static_assert((gemm_stages(BM, BN, WM, WN) - 1) * ((BM + BN) / 8 / (WM * WN)) <= 63, "vmcnt is 6 bits");
    a.stage_out = ((a.ldy % 8) == 0 && (((uintptr_t)a.Y) & 15) == 0) ? 1 : 0;
#define VDD_GEMM_LAUNCH(E)                                                                                            \
    case E: {                                                                                                         \
        auto kfn = gemm_kernel<BM, BN, WM, WN, E>;                                                                       \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; } \
        hipLaunchKernelGGL(kfn, grid, block, smem, st, a);                                                            \
        break;                                                                                                        \
    }
    if (a.slab_S > 0) {
        if constexpr (BM == 64) { epi = EPI_SLABS; switch (epi) { VDD_GEMM_LAUNCH(EPI_SLABS) } return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH; }
        else return VDD_ERR_INVALID_ARG;           // split-K slabs: the 64 x 256 tile (config 8) only
    }
    switch (epi) {
        VDD_GEMM_LAUNCH(EPI_NONE)
        VDD_GEMM_LAUNCH(EPI_BIAS)
        VDD_GEMM_LAUNCH(EPI_BIAS_QUICK_GELU)
        VDD_GEMM_LAUNCH(EPI_BIAS_GELU)
        VDD_GEMM_LAUNCH(EPI_BIAS_RESID)
        case EPI_SWIGLU:
            if constexpr ((BN / WN / 32) % 2 == 0 && 256 % BN == 0) {           // gate / up pairs inside a wave; F % 128 == 0 tiles the features
                switch (epi) { VDD_GEMM_LAUNCH(EPI_SWIGLU) }
                break;
            } else {
                return VDD_ERR_INVALID_ARG;          // this tile shape cannot pair gate / up blocks inside a wave
            }
        default: return VDD_ERR_INVALID_ARG;
    }
#undef VDD_GEMM_LAUNCH
    return hipGetLastError() == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH;
}

}  // namespace VDD_ELEM_NS
}  // namespace

using namespace VDD_ELEM_NS;

extern "C" {

VDD_HIDDEN int64_t VDD_IMPL(vdd_gemm_workspace_bytes)(int M, int N) {
    // fixed counter region (up to 2^20 tiles per launch) + one 256 x 256 fp32 partial per workgroup; the same for every shape
    (void)M; (void)N;
    return (int64_t)COUNTER_BYTES + (int64_t)num_workgroups() * 256 * 256 * 4;
}

VDD_HIDDEN int VDD_IMPL(vdd_gemm)(const void* X, const void* W, void* Y, const void* bias, const void* resid, int M, int N, int K,
             int64_t ldx, int64_t ldw, int64_t ldy, int64_t ldr, int epilogue, int config, void* workspace, int64_t workspace_bytes,
             void* stream) {
    if (M <= 0 || N <= 0) return VDD_OK;
    if (!X || !W || !Y || K <= 0 || K % 128 != 0 || (ldx % 8) || (ldw % 8) || (ldy % 4) || N % 4 != 0) return VDD_ERR_INVALID_ARG;
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_QUICK_GELU || epilogue == EPI_BIAS_GELU || epilogue == EPI_BIAS_RESID) && !bias) return VDD_ERR_INVALID_ARG;
    if (epilogue == EPI_BIAS_RESID && (!resid || (ldr % 4))) return VDD_ERR_INVALID_ARG;
    if (epilogue == EPI_SWIGLU && N % 128 != 0) return VDD_ERR_INVALID_ARG;
    // per-lane byte offsets inside a tile are 32-bit: 256 rows of X, and for SwiGLU the gate -> up distance in W
    if ((uint64_t)256 * (uint64_t)ldx * 2 >= 0x7fffffffull) return VDD_ERR_INVALID_ARG;
    if ((uint64_t)((epilogue == EPI_SWIGLU ? (uint64_t)N : 0) + 256) * (uint64_t)ldw * 2 >= 0x7fffffffull) return VDD_ERR_INVALID_ARG;
    GemmArgs a{};
    a.X = (const uint16_t*)X; a.W = (const uint16_t*)W; a.Y = (uint16_t*)Y;
    a.bias = (const uint16_t*)bias; a.resid = (const uint16_t*)resid;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr;
    hipStream_t st = (hipStream_t)stream;
    const int sched = (config >> 4) & 3;            // tuning: 0 hybrid, 1 data-parallel only, 2 stream-K only, 3 split-K slabs
    if (sched == 3) {                               // Y = fp32 [S][M][N] (ldy = N), S = config bits 8-15; plain product only
        a.slab_S = (config >> 8) & 255;
        a.slabs = (float*)Y;
        if (a.slab_S < 1 || epilogue != EPI_NONE || ldy != N || ((uintptr_t)Y & 15)) return VDD_ERR_INVALID_ARG;
    }
    config = (config & 15) | (((config >> 6) & 3) << 4);      // macro tile id: bits 0-3 + bits 6-7 (ids 16 ..: round 6)
    if (config == 0) config = 1;
    switch (config) {
        case 1: return launch_cfg<256, 256, 2, 4>(a, epilogue, sched, workspace, workspace_bytes, st);
        case 2: return launch_cfg<128, 256, 2, 4>(a, epilogue, sched, workspace, workspace_bytes, st);
        case 3: return launch_cfg<256, 128, 4, 2>(a, epilogue, sched, workspace, workspace_bytes, st);
        case 4: return launch_cfg<192, 256, 2, 4>(a, epilogue, sched, workspace, workspace_bytes, st);
        case 5: return launch_cfg<256, 192, 4, 2>(a, epilogue, sched, workspace, workspace_bytes, st);
        case 6: return launch_cfg<192, 192, 2, 3>(a, epilogue, sched, workspace, workspace_bytes, st);      // 6 waves of 96 x 64
        case 7: return launch_cfg<192, 128, 2, 4>(a, epilogue, sched, workspace, workspace_bytes, st);      // 8 waves of 96 x 32
        case 8: return launch_cfg<64, 256, 2, 4>(a, epilogue, sched, workspace, workspace_bytes, st);       // a few dozen rows: W streaming,
                                                                                                            // 64-KiB partial slabs
        // W streams for 65 - 256 rows (gemm_stages: 4 - 5 K-tile buffers); 10 / 11 pair gate / up inside a wave (SwiGLU epilogue)
        case 9: return launch_cfg<128, 128, 2, 4>(a, epilogue, sched, workspace, workspace_bytes, st);      // 8 waves of 64 x 32
        case 10: return launch_cfg<128, 128, 2, 2>(a, epilogue, sched, workspace, workspace_bytes, st);     // 4 waves of 64 x 64
        case 11: return launch_cfg<192, 128, 2, 2>(a, epilogue, sched, workspace, workspace_bytes, st);     // 4 waves of 96 x 64
        // up to 64 rows (round 6): the row tile is the batch, six K-tile buffers; 12 pairs gate / up inside a wave
        case 12: return launch_cfg<64, 128, 2, 2>(a, epilogue, sched, workspace, workspace_bytes, st);      // 4 waves of 32 x 64
        case 13: return launch_cfg<64, 128, 2, 4>(a, epilogue, sched, workspace, workspace_bytes, st);      // 8 waves of 32 x 32
        case 14: return launch_cfg<32, 128, 1, 4>(a, epilogue, sched, workspace, workspace_bytes, st);      // up to 32 rows: 4 waves of 32 x 32, eight buffers
        case 15: return launch_cfg<32, 128, 1, 2>(a, epilogue, sched, workspace, workspace_bytes, st);      // 2 waves of 32 x 64 (gate / up pairs)
        // a 96-row tile: a batch of 270 rows (90 questions x 3 branches, BASELINE config #3) is three of them (288 rows) where 128- / 192-row tiles compute
        // 384.  Measured (tools/gemm_96_tile_ab.py): the tuner takes it for the d x d attention-output projection at 90 - 360 rows (13B step at 270 rows
        // 22.8 -> 22.0 ms) and for nothing else - the wide projections keep their 192-row tiles, whose padding costs less than streaming W a third
        // time; the 2-wave and 96 x 256 variants of it were never picked and are not built
        case 16: return launch_cfg<96, 128, 1, 4>(a, epilogue, sched, workspace, workspace_bytes, st);      // 4 waves of 96 x 32, five buffers
        // (launch_cfg<256, 256, 2, 2> - four waves of 128 x 128, one per SIMD with 512 registers, the vendor kernel's shape - instantiates as it
        //  is and was measured: 1.10 PF/s on random data, 1.27 on zero operands against 1.27 / 1.6 for the 8-wave tile: a lone wave per SIMD
        //  stalls its own MFMAs behind every LDS-DMA issue and fragment read.  profiles/r04_gemm_data_dependence.jsonl)
        default: return VDD_ERR_INVALID_ARG;
    }
}

}  // extern "C"
