// Weight-streaming projections for 17 - 64 rows in flight, K cut over workgroups (gfx950 / MI355X).
//
// The regime: a few dozen (question, branch) rows per decode step - what one rank of an 8-GPU split of BASELINE config #3 / a 4-GPU
// split of #5 holds (experiments/eval/MME/run_llava.py:32-40 `get_chunk`), or a handful of questions through
// experiments/eval/calibrate/llava_calibrate.py:161-177.  Every projection of HF's LlamaDecoderLayer [ext] under
// experiments/llava/model/language_model/llava_llama.py:88-103 is then a stream of W through the chip with almost no arithmetic,
// and what the round-4 kernels paid for was X: 16 / 32-column blocks re-read ALL of X (64 x K) per block from L2 - 2 - 4 X
// fragments per W fragment, 200 - 400 MB of L2 traffic for 100 MB of weights - and the MFMA GEMM's 64 x 256 tiles cover a fraction of
// the CUs and end in a serial stream-K fix-up.
//
// Here the grid is `teams x KS` workgroups (= the CU count).  Workgroup (team, ks) owns a contiguous range of 16-column tiles of W
// and the ks-th slab of K:
//   * it stages X[:, slab] ONCE into LDS (M x 2 slab bytes <= 150 KiB; swizzled 16-byte chunks, conflict-free ds_read_b128) - 32 MB of
//     L2 reads for the qkv projection at 64 rows instead of 400.  NORM: X is the un-normalised residual stream H and the staging
//     applies bf16(bf16(h * rstd) * ln_w[k]) (the roundings of rmsnorm_kernel), rstd from the producer's partial sums of squares: the
//     layer's RMSNorm launches disappear as they do below 17 rows;
//   * each of its 4 waves walks its own column tiles: W fragments go global -> registers (every byte of W is read by one wave, once),
//     four register stages of eight k-steps run ahead ACROSS tile boundaries (16 - 24 KiB in flight per wave: the 2 us x 6 TB/s the
//     memory system needs), MFMA 16x16x32 against the X fragments in LDS;
//   * a finished (tile, slab) partial goes to an fp32 slab with write-through stores; the team's KS workgroups then MEET at one
//     arrival counter and each runs the epilogue of every KS-th tile of the team, adding the slabs in slab order (deterministic):
//     bf16 round (+ residual, + per-row sums of squares of the tile's 16 columns for the next NORM staging), or SiLU(gate) * up
//     for W = [Wg; Wu].  (The first form - the LAST arriver of a tile does its epilogue, nobody waits - left all of a team's tiles
//     to its one late workgroup: 6 - 18 us tails.)
// The meeting ASSUMES the whole grid (one workgroup per CU) is co-resident: under a CU mask, beside another stream's kernel or on a
// partitioned part a team's workgroups may never all run at once.  The wait is therefore bounded, and a workgroup that gives up writes
// 1 + its index into the workspace's status word (vdd_skinny_slab_status_offset) before it goes on: the launch's outputs are garbage
// then, the C entry still returns VDD_OK (it is asynchronous), and the caller reads the word after a sync (lost_ops.slab_status).
// Laboratory code (round 6): measured 1.07 - 1.25x SLOWER than the shipped 17 - 64-row path, not part of libvdd_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdd_elem.h"

namespace {
namespace VDD_ELEM_NS {
using namespace vdd_elem;

typedef __attribute__((ext_vector_type(8))) short frag8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

struct SlabArgs {
    const uint16_t* X; const float* ss; const uint16_t* lnw; const uint16_t* W; const uint16_t* R; uint16_t* Y; float* ss_out;
    float* part; int* tickets;
    long long ldx, ldr, ldy;
    int M, N, K, NT, KS, teams, TK, RS, nss;
    float eps;
#ifdef VDD_PROBE_BUILD
    long long* dbg;                     // tools/slab_timeline.py: per-wave phase timestamps
#endif
};

#ifdef VDD_PROBE_BUILD
#define SLAB_TS(v) do { if (a.dbg) v = wall_clock64(); } while (0)
#define SLAB_TS_ONCE(v) do { if (a.dbg && v == 0) v = wall_clock64(); } while (0)
#else
#define SLAB_TS(v) do { } while (0)
#define SLAB_TS_ONCE(v) do { } while (0)
#endif

constexpr int SLAB_NW = 4;            // waves per workgroup (one per SIMD: the register file holds the four-stage W pipeline)
constexpr int SLAB_STAGES = 4;
constexpr int SLAB_LDS_CAP = 150 * 1024;
constexpr int SLAB_MAX_TILES = 16384;  // ticket words at the head of the workspace (N <= 262,144 output columns)
constexpr int SLAB_STATUS_WORD = SLAB_MAX_TILES - 1;   // the give-up word (team counters use words < 2 x the CU count)

__device__ __forceinline__ uint32_t norm_pair(uint32_t hv, uint32_t gv, float rstd) {
    const uint32_t nb = cvt_pk(lo(hv) * rstd, hi(hv) * rstd);
    return cvt_pk(lo(nb) * lo(gv), hi(nb) * hi(gv));
}

template <int MT, bool SWIGLU>
__global__ void __launch_bounds__(SLAB_NW * 64) skinny_slab_kernel(const SlabArgs a) {
    constexpr int C = SWIGLU ? 2 : 1;               // W fragments per k-step (gate + up rows of the same features)
    constexpr int U = SWIGLU ? 4 : 8;               // k-steps (32 elements each) per register stage: 8 KiB of W per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_lds[];
    float* rstd_s = reinterpret_cast<float*>(slab_lds);                    // [64]
    unsigned char* xs = slab_lds + 512;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform for the compiler too (SGPR descriptors)
    const int ln = lane & 15, g = lane >> 4;
    const int team = blockIdx.x / a.KS, ks = blockIdx.x - team * a.KS;
    const int c0 = (int)((long long)team * a.NT / a.teams), c1 = (int)((long long)(team + 1) * a.NT / a.teams);
    const int s0 = (int)((long long)ks * a.TK / a.KS), s1 = (int)((long long)(ks + 1) * a.TK / a.KS);
    const int nks = s1 - s0, kbeg = s0 * 32;
    const int nt = (c1 - c0 - wave + SLAB_NW - 1) / SLAB_NW > 0 ? (c1 - c0 - wave + SLAB_NW - 1) / SLAB_NW : 0;     // this wave's tiles
    const int nb = (nks + U - 1) / U, TB = nt * nb;
    [[maybe_unused]] long long t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0, t4_ = 0, t5_ = 0, t6_ = 0;
    SLAB_TS(t0_);

    // ---- stage X[:, slab] into LDS, FIRST and alone: a wave's loads return in order, and behind the W prefetch (every wave of the
    // chip fires 24 KiB at once, a 25 MB burst the memory system takes 4 - 5 us to serve) X arrived 7 us into the launch
    const bool norm = a.ss != nullptr;
    const int cpr = nks * 4;                                                // 16-byte chunks of a row in this slab
    constexpr int RB = 64 / SLAB_NW;                                        // rows per wave
    float4 sv[16];
    const int nr = tid >> 2, npart = tid & 3, nper4 = a.nss >> 4;           // rstd: four lanes per row, a quarter of the partials each
    if (norm && nr < a.M) {
        const float4* p = reinterpret_cast<const float4*>(a.ss + (size_t)nr * a.nss + npart * (a.nss >> 2));
#pragma unroll
        for (int j = 0; j < 16; ++j) sv[j] = j < nper4 ? p[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int q0 = 0; q0 < cpr; q0 += 128) {
        uint4 xv[2][RB];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int q = q0 + qi * 64 + lane, r = wave + j * SLAB_NW;
                xv[qi][j] = make_uint4(0, 0, 0, 0);
                if (q < cpr && r < a.M) xv[qi][j] = *reinterpret_cast<const uint4*>(a.X + (size_t)r * a.ldx + kbeg + q * 8);
            }
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int q = q0 + qi * 64 + lane, r = wave + j * SLAB_NW;
                if (q < cpr && r < a.M) *reinterpret_cast<uint4*>(xs + (size_t)r * a.RS + ((q ^ (r & 15)) << 4)) = xv[qi][j];
            }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the W pipeline: batch i = (tile j = i / nb, k-batch i % nb); loads run SLAB_STAGES - 1 batches ahead of the MFMAs
    frag8_t wb[SLAB_STAGES][U][C];
    int lj = 0, lkb = 0;
    auto ldb = [&](frag8_t (&b)[U][C]) {
        int row = (c0 + wave + lj * SLAB_NW) * 16 + ln;
        if (row >= a.N) row = a.N - 1;
        const uint16_t* p0 = a.W + (size_t)row * a.K + kbeg + g * 8;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int s = lkb * U + u; if (s >= nks) s = nks - 1;                 // beyond the slab: a valid address, the fragment is zeroed at use
            b[u][0] = *reinterpret_cast<const frag8_t*>(p0 + s * 32);
            if constexpr (SWIGLU) b[u][1] = *reinterpret_cast<const frag8_t*>(p0 + (size_t)a.N * a.K + s * 32);
        }
        if (++lkb == nb) { lkb = 0; ++lj; }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int st = 0; st < SLAB_STAGES - 1; ++st) if (st < TB) ldb(wb[st]);

    if (norm) {
        // rstd: in index order per lane, then a fixed tree over the four lanes of a row - every workgroup computes the same bits
        float ssum = 0.f;
        if (nr < a.M) {
#pragma unroll
            for (int j = 0; j < 16; ++j) ssum += (sv[j].x + sv[j].y) + (sv[j].z + sv[j].w);
            const float4* p = reinterpret_cast<const float4*>(a.ss + (size_t)nr * a.nss + npart * (a.nss >> 2));
            for (int j = 16; j < nper4; ++j) { const float4 v = p[j]; ssum += (v.x + v.y) + (v.z + v.w); }      // nss > 256 partials per row
        }
        const float s1_ = __shfl_xor(ssum, 1);
        const float pair = (npart & 1) ? s1_ + ssum : ssum + s1_;           // (q0 + q1), (q2 + q3): the same operands in the same order on both lanes
        const float p2 = __shfl_xor(pair, 2);
        const float tot = (npart & 2) ? p2 + pair : pair + p2;
        if (npart == 0 && nr < 64) rstd_s[nr] = nr < a.M ? rsqrtf(tot / (float)a.K + a.eps) : 0.f;
        __syncthreads();
        // normalise the image in place: bf16(bf16(h * rstd) * ln_w[k]), the roundings of rmsnorm_kernel (one small loop: the kernel's
        // cold code is fetched once per CU pair and every KiB of it is paid in the prologue)
        for (int q = lane; q < cpr; q += 64) {
            const uint4 gw = *reinterpret_cast<const uint4*>(a.lnw + kbeg + q * 8);
            for (int r0 = wave; r0 < a.M; r0 += 4 * SLAB_NW) {
                uint4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = r0 + j * SLAB_NW < a.M ? r0 + j * SLAB_NW : r0;
                    v[j] = *reinterpret_cast<const uint4*>(xs + (size_t)r * a.RS + ((q ^ (r & 15)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = r0 + j * SLAB_NW;
                    if (r < a.M) {
                        const float rs = rstd_s[r];
                        *reinterpret_cast<uint4*>(xs + (size_t)r * a.RS + ((q ^ (r & 15)) << 4)) =
                            make_uint4(norm_pair(v[j].x, gw.x, rs), norm_pair(v[j].y, gw.y, rs), norm_pair(v[j].z, gw.z, rs), norm_pair(v[j].w, gw.w, rs));
                    }
                }
            }
        }
    }
    __syncthreads();
    SLAB_TS(t1_);

    // ---- main loop
    int xrow[MT], xsw[MT];                      // LDS row base / swizzle key of this lane's row in M-tile t (rows beyond M read row M - 1)
#pragma unroll
    for (int t = 0; t < MT; ++t) { int r = t * 16 + ln; if (r >= a.M) r = a.M - 1; xrow[t] = r * a.RS; xsw[t] = r & 15; }
    f32x4_t acc[MT][C];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int cj = 0, ckb = 0;
    auto part_rsrc = [&](int tile) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.part + (size_t)tile * a.KS * (MT * C * 256)), 0, a.KS * MT * C * 1024, 0x00020000);
    };
    auto flush = [&]() {
        // write-through (sc1) stores: the partial leaves this XCD's L2 at once, no release fence (vdd_gemm.hip's slab exchange)
        const int tile = c0 + wave + cj * SLAB_NW;
        const __amdgpu_buffer_rsrc_t rp = part_rsrc(tile);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[t][c]), rp, ((ks * MT + t) * C + c) * 1024 + lane * 16, 0, 16);
                acc[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
    };
    auto mmb = [&](const frag8_t (&b)[U][C]) {
        const frag8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = ckb * U + u;
            const bool valid = s < nks;
            const int q = (valid ? s : nks - 1) * 4 + g;
            frag8_t xf[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) xf[t] = *reinterpret_cast<const frag8_t*>(xs + xrow[t] + ((q ^ xsw[t]) << 4));
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const frag8_t bw = valid ? b[u][c] : zero;
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t][c] = mfma16(xf[t], bw, acc[t][c]);
            }
        }
        if (++ckb == nb) { flush(); ckb = 0; ++cj; }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < TB; i += SLAB_STAGES) {
#pragma unroll
        for (int st = 0; st < SLAB_STAGES; ++st) {
            if (i + st + SLAB_STAGES - 1 < TB) ldb(wb[(st + SLAB_STAGES - 1) % SLAB_STAGES]);
            if (i + st < TB) mmb(wb[st]);
        }
    }
    SLAB_TS(t2_);

    // ---- the team's KS workgroups meet (arrival counter), then each finishes ITS share of the team's tiles - every KS-th one - so
    // the epilogue work is spread over all of them (a last-arriver rule gave all of a team's tiles to the one workgroup that was
    // late: 3 tiles per wave, 6 - 18 us of tail).  The wait assumes the grid is co-resident: it is at most one workgroup per CU.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // every wave's partials have landed; the X image is scratch from here on
    int* cnt = a.tickets + team * 2;
    if (tid == 0) {
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // bounded (~seconds): counters left dirty by a launch that died must not hang the device; the results are garbage then and
        // the caller re-zeroes the workspace (ops.slab_workspace_reset)
        int spin = 0;
        for (; spin < (1 << 22) && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.KS; ++spin) __builtin_amdgcn_s_sleep(2);
        if (spin == (1 << 22)) {                     // gave up: say so (first one wins) - the sums below read slabs nobody wrote
            int zero = 0;
            __hip_atomic_compare_exchange_strong(a.tickets + SLAB_STATUS_WORD, &zero, 1 + (int)blockIdx.x, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
    SLAB_TS(t3_);
    float* sums = reinterpret_cast<float*>(xs) + wave * ((MT * C + MT) * 256);          // per wave: the tile's sums [MT][C][64][4] + residual [MT][64][4]
    float* rres = sums + MT * C * 256;
    for (int i = ks + wave * a.KS; c0 + i < c1; i += SLAB_NW * a.KS) {
        const int tile = c0 + i, col = tile * 16 + ln;
        // the residual entries and the slabs (eight at a time) go out together: one memory round trip; added in slab order
        float rv[MT][4];
        if (!SWIGLU && a.R != nullptr) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = t * 16 + g * 4 + e;
                    rv[t][e] = row < a.M && col < a.N ? e2f(a.R[(size_t)row * a.ldr + col]) : 0.f;
                }
        }
        const __amdgpu_buffer_rsrc_t rq = part_rsrc(tile);
        f32x4_t tot[MT][C];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) tot[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        constexpr int KCH = 8;
        for (int k0 = 0; k0 < a.KS; k0 += KCH) {
            f32x4_t v[KCH][MT][C];
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                if (k0 + k >= a.KS) continue;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        v[k][t][c] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, (((k0 + k) * MT + t) * C + c) * 1024 + lane * 16, 0, 16));
            }
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                if (k0 + k >= a.KS) continue;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int c = 0; c < C; ++c) { tot[t][c][0] += v[k][t][c][0]; tot[t][c][1] += v[k][t][c][1]; tot[t][c][2] += v[k][t][c][2]; tot[t][c][3] += v[k][t][c][3]; }
            }
        }
        // through LDS, so that the epilogue is ONE small run-time loop over (M tile, entry) instead of MT x 4 unrolled copies
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4_t*>(sums + (t * C + c) * 256 + lane * 4) = tot[t][c];
            if (!SWIGLU && a.R != nullptr) *reinterpret_cast<f32x4_t*>(rres + t * 256 + lane * 4) = f32x4_t{rv[t][0], rv[t][1], rv[t][2], rv[t][3]};
        }
        SLAB_TS_ONCE(t5_);
        // C/D map of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + e
        for (int te = 0; te < MT * 4; ++te) {
            const int t = te >> 2, e = te & 3, row = t * 16 + g * 4 + e;
            if (t * 16 >= a.M) break;
            float sq = 0.f;
            if (row < a.M && col < a.N) {
                uint32_t ob;
                if constexpr (SWIGLU) {
                    const float gb = e2f(f2e(sums[(t * C + 0) * 256 + lane * 4 + e])), ub = e2f(f2e(sums[(t * C + 1) * 256 + lane * 4 + e]));
                    const float sl = e2f(f2e(gb / (1.f + __expf(-gb))));
                    ob = f2e(sl * ub);
                } else {
                    float o = e2f(f2e(sums[t * 256 + lane * 4 + e]));
                    if (a.R != nullptr) o = o + rres[t * 256 + lane * 4 + e];
                    ob = f2e(o);
                }
                a.Y[(size_t)row * a.ldy + col] = (uint16_t)ob;
                const float h = e2f(ob);
                sq = h * h;
            }
            if (a.ss_out != nullptr) {       // the 16 lanes ln = 0..15 of a lane group hold the tile's 16 columns of one row
                sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4); sq += __shfl_xor(sq, 8);
                if (ln == 0 && row < a.M) a.ss_out[(size_t)row * a.NT + tile] = sq;
            }
        }
        SLAB_TS_ONCE(t6_);
    }
    // the last workgroup of the team to leave zeroes the counters for the next launch
    if (tid == 0 && __hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.KS - 1) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef VDD_PROBE_BUILD
    if (a.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t4_ = wall_clock64();
        if (lane == 0) { long long* d = a.dbg + ((size_t)blockIdx.x * SLAB_NW + wave) * 8; d[0] = t0_; d[1] = t1_; d[2] = t2_; d[3] = t3_; d[4] = t4_; d[5] = t5_; d[6] = t6_; d[7] = nt; }
    }
#endif
}

#ifdef VDD_PROBE_BUILD
static long long* g_slab_dbg = nullptr;
#endif
// ---- host side: the cut of one projection over the chip
struct SlabPlan { int MT, KS, teams, NT, TK, RS, lds; int64_t ws_bytes; int64_t tickets_bytes; };

static int n_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) n = 256;
        else n = p.multiProcessorCount;
    }
    return n;
}

// false: this shape is not served (the caller takes the other weight-streaming kernels / the GEMM)
static bool slab_plan(int M, int N, int K, int swiglu, SlabPlan& p) {
    if (M < 1 || M > 64 || N < 16 || K < 256 || K % 32 != 0) return false;
    p.MT = (M + 15) / 16;
    p.NT = (N + 15) / 16;
    p.TK = K / 32;
    const int cus = n_cus() >= 32 ? n_cus() / 32 * 32 : 32;
    int kslab_max = SLAB_LDS_CAP / (2 * M) / 128 * 128;             // elements of a row the LDS image may hold
    if (kslab_max < 128) return false;
    // KS: the power of two (a slab fits the LDS, at least four k-steps deep) with the shortest critical path - k-steps of the wave
    // that owns the most tiles - plus the partial slabs' traffic priced in k-steps of the weight stream (2 KS M fp32 per W element
    // pair, written and read); ties go to the smaller cut
    int ks = 0;
    double best = 0.0;
    const double ideal = (double)p.NT * p.TK / (cus * SLAB_NW);
    for (int c = 1; c <= 32; c *= 2) {
        const int kst = (p.TK + c - 1) / c;
        if (kst * 32 > kslab_max || kst < 4) continue;
        const int teams = cus / c < p.NT ? cus / c : p.NT;
        const int tiles = (p.NT + teams - 1) / teams;
        const double cost = (double)((tiles + SLAB_NW - 1) / SLAB_NW) * kst + (c > 1 ? ideal * 2.0 * c * M / K : 0.0);
        if (ks == 0 || cost < best - 1e-9) { ks = c; best = cost; }
    }
    if (ks == 0) return false;
    p.KS = ks;
    p.teams = cus / ks < p.NT ? cus / ks : p.NT;
    const int kslab = (p.TK + ks - 1) / ks * 32;
    p.RS = (kslab + 127) / 128 * 256;
    p.lds = 512 + (M * p.RS > 49152 ? M * p.RS : 49152);             // >= the epilogue's scratch (<= 12 KiB per wave)
    const int C = swiglu ? 2 : 1;
    if (p.NT > SLAB_MAX_TILES || 2 * cus >= SLAB_STATUS_WORD) return false;
    p.tickets_bytes = (int64_t)SLAB_MAX_TILES * 4;                    // a FIXED region: launches of different widths share one workspace
    p.ws_bytes = p.tickets_bytes + (int64_t)p.NT * ks * p.MT * C * 1024;
    return true;
}

template <int MT, bool SWIGLU>
static hipError_t slab_launch_mt(const SlabArgs& a, const SlabPlan& p, hipStream_t st) {
    static int attr = 0;
    if (attr < p.lds) {
        hipError_t e = hipFuncSetAttribute((const void*)skinny_slab_kernel<MT, SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, 512 + SLAB_LDS_CAP);
        if (e != hipSuccess) return e;
        attr = 512 + SLAB_LDS_CAP;
    }
    hipLaunchKernelGGL((skinny_slab_kernel<MT, SWIGLU>), dim3(p.teams * p.KS), dim3(SLAB_NW * 64), p.lds, st, a);
    return hipGetLastError();
}

}  // namespace VDD_ELEM_NS
}  // namespace

using namespace VDD_ELEM_NS;

extern "C" {

#ifdef VDD_PROBE_BUILD
__attribute__((visibility("default"))) void VDD_IMPL(vdd_dbg_slab_timeline)(void* p) { g_slab_dbg = (long long*)p; }
#endif

VDD_HIDDEN int64_t VDD_IMPL(vdd_skinny_slab_status_offset)(void) { return (int64_t)SLAB_STATUS_WORD * 4; }

VDD_HIDDEN int64_t VDD_IMPL(vdd_skinny_slab_workspace_bytes)(int M, int N, int K, int swiglu) {
    SlabPlan p;
    return slab_plan(M, N, K, swiglu, p) ? p.ws_bytes : -1;
}

VDD_HIDDEN int VDD_IMPL(vdd_skinny_slab)(const void* X, const float* ss, int nss, const void* ln_w, float eps, const void* W, const void* R, void* Y,
                                         float* ss_out, int M, int N, int K, int64_t ldx, int64_t ldr, int64_t ldy, int swiglu, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0) return VDD_OK;
    SlabPlan p;
    if (!X || !W || !Y || !workspace || (ldx % 8) != 0 || (ss && (!ln_w || nss < 16 || nss % 16 != 0)) || (swiglu && (R || ss_out)))
        return VDD_ERR_INVALID_ARG;
    if (!slab_plan(M, N, K, swiglu, p)) return VDD_ERR_UNSUPPORTED;
    if (workspace_bytes < p.ws_bytes) return VDD_ERR_INVALID_ARG;
    SlabArgs a;
    a.X = (const uint16_t*)X; a.ss = ss; a.lnw = (const uint16_t*)ln_w; a.W = (const uint16_t*)W; a.R = (const uint16_t*)R; a.Y = (uint16_t*)Y;
    a.ss_out = ss_out; a.tickets = (int*)workspace; a.part = (float*)((char*)workspace + p.tickets_bytes);
    a.ldx = ldx; a.ldr = ldr; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.NT = p.NT; a.KS = p.KS; a.teams = p.teams; a.TK = p.TK; a.RS = p.RS;
    a.nss = nss; a.eps = eps;
#ifdef VDD_PROBE_BUILD
    a.dbg = g_slab_dbg;
#endif
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (swiglu) e = p.MT == 1 ? slab_launch_mt<1, true>(a, p, st) : p.MT == 2 ? slab_launch_mt<2, true>(a, p, st) : p.MT == 3 ? slab_launch_mt<3, true>(a, p, st) : slab_launch_mt<4, true>(a, p, st);
    else e = p.MT == 1 ? slab_launch_mt<1, false>(a, p, st) : p.MT == 2 ? slab_launch_mt<2, false>(a, p, st) : p.MT == 3 ? slab_launch_mt<3, false>(a, p, st) : slab_launch_mt<4, false>(a, p, st);
    return e == hipSuccess ? VDD_OK : VDD_ERR_LAUNCH;
}

}  // extern "C"
