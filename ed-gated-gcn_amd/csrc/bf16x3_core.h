// Shared main loop of the bf16x3 matrix kernels (linear_split.hip, fused_layer.hip).
//
//   acc[128 x 64 per wavefront] += A[128 rows, K] (fp32, split on the fly) . Wpack[K, 256 cols]
//
// fp32 operands are split into two bf16 terms, x = hi + lo (hi = RNE bf16(x), lo = RNE
// bf16(x - hi), residual <= 2^-16 |x|); each product is three bf16 MFMAs with fp32 accumulation:
//     x.w ~= hi.hi + lo.hi + hi.lo            (dropped lo.lo <= 2^-16 |x.w|)
//
// Geometry (MI355X): workgroup = 256 threads = 4 wavefronts side by side along F, tile
// 128 (rows) x 256 (columns), two workgroups per CU (2 wavefronts per SIMD, from DIFFERENT
// workgroups, so one's split/LDS-write/barrier phase overlaps the other's MFMAs).  Each
// wavefront owns 128 x 64 = 4 x 2 MFMA tiles of 32x32 (128 accumulator VGPRs).
//   * A: 16-B global loads (one 128-B line per 8 lanes), split in registers
//     (v_cvt_pk_bf16_f32) BETWEEN the MFMAs of the previous stage, two bf16 planes in LDS,
//     double-buffered: ONE barrier per 32-deep stage.  LDS rows are 64 B; the 16-B chunk index is XORed with (row>>2)&3 so every
//     ds_read_b128 lane group touches 16 distinct 16-B slots (measured: 0 bank conflicts).
//   * W never touches LDS: ggcn_weight_pack stores it once in MFMA B-fragment order
//     [n_tile][k_step][hi|lo][lane][8 x bf16]; a fragment is one coalesced 1 KiB load from L2,
//     issued one k-step ahead.
//   * loads are unconditional (a predicated load makes hipcc drain vmcnt at the join) and are
//     pinned in issue order with sched_barrier: hipcc otherwise sinks them next to their first
//     use and exposes the whole HBM/L2 latency every stage (measured 470 -> 400 us).
//
// Operand lane maps of v_mfma_f32_32x32x16_bf16 (cdna guide §3): lane l, r = l&31, h = l>>5:
// A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7; C/D: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#pragma once
#include "common.h"

#include <hip/hip_fp16.h>

#include <type_traits>

namespace ggcn {
namespace bx3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// GGCN_RN = 32-column MFMA tiles per wavefront: 2 (default: 4 wavefronts side by side, 128x64 each,
// two workgroups per CU) or 4 (2 x 2 wavefronts, 128x128 each, 256 accumulator registers, ONE
// workgroup per CU = one wavefront per SIMD with the whole 512-register file).
#ifndef GGCN_RN
#define GGCN_RN 2
#endif
constexpr int RN = GGCN_RN;
constexpr int WN = 8 / RN;            // wavefronts along F
constexpr int WM = 4 / WN;            // wavefronts along the rows
constexpr int kWavesPerSimd = (RN == 2) ? 2 : 1;
constexpr int BM = 128 * WM, BN = 256, BK = 32;
constexpr int KSTEP = 16;             // K per MFMA
constexpr int KS = BK / KSTEP;        // MFMA k-steps per stage
constexpr int ROWB = BK * 2;          // bytes per LDS row of one plane (64)
constexpr int NT = 32;                // columns per MFMA tile
constexpr int FRAG_BYTES = 64 * 16;   // one B fragment: 64 lanes x 8 bf16
constexpr int kThreads = 256;
constexpr int kLdsBytes = 2 * 2 * BM * ROWB;  // [buffer][plane][BM rows x 64 B] = 32 KiB (64 KiB when BM = 256)

// Staging geometry of one 128 x 32 stage of A for element type AT: every thread moves 16 B per
// pass.  fp32: 8 threads per row, 32 rows per pass, 4 passes;  fp16: 4 per row, 64 rows, 2 passes.
template <typename AT>
struct Geom {
    static constexpr int EPT = 16 / (int)sizeof(AT);  // elements per thread per pass
    static constexpr int TPR = BK / EPT;              // threads per row
    static constexpr int RPP = kThreads / TPR;        // rows per pass
    static constexpr int NP = BM / RPP;               // passes
};

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// LDS image of one A plane: [128 rows][4 chunks of 16 B]; the chunk index is XORed with
// (row>>2)&3 so that the 16 rows of a ds_read_b128 lane group land on 16 distinct 16-B slots of
// the 256-B bank row (4 rows of 64 B per bank row).
__device__ __forceinline__ int a_lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }
// staging row of pass p for this thread
template <typename AT>
__device__ __forceinline__ int stage_row(int p) { return p * Geom<AT>::RPP + (int)(threadIdx.x & (kThreads - 1)) / Geom<AT>::TPR; }

// Block id -> tile.  Observed dispatch (speed only, never correctness): ids round-robin over the
// 8 XCDs; inside an XCD the first 32 blocks take the 32 CUs in order, the next 32 become their
// co-residents, and a finished block is replaced by the block 64 slots later.
//   * the column tiles of one row block sit on neighbouring CUs of ONE XCD at the same time, so
//     they share that row block's A rows through the XCD's L2.
__device__ __forceinline__ bool tile_of_block(int id, int m_tiles, int n_wg, int &m_tile, int &n_wgi)
{
    const int xcd = id & 7, slot = id >> 3;
    m_tile = (slot / n_wg) * 8 + xcd;
    n_wgi = slot % n_wg;
    return m_tile < m_tiles;
}
inline int64_t grid_for(int64_t m_tiles, int n_wg)
{
    const int64_t per_xcd = (m_tiles + 7) / 8;          // row blocks per XCD
    return per_xcd * n_wg * 8;
}

__device__ __forceinline__ float elem_to_float(float v) { return v; }
__device__ __forceinline__ float elem_to_float(__half v) { return __half2float(v); }

// 16 B of AT from global memory as EPT floats
template <typename AT>
__device__ __forceinline__ void load16(const AT *p, float (&v)[Geom<AT>::EPT])
{
    if constexpr (sizeof(AT) == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        const __half2 *h = reinterpret_cast<const __half2 *>(&t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 f = __half22float2(h[i]);
            v[2 * i] = f.x;
            v[2 * i + 1] = f.y;
        }
    }
}

// arow[i]: this thread's NP source rows (already clamped to valid memory); avalid[i]: false =>
// the row is padding and must read as zeros.  AT: element type of A (float or __half: an fp16
// value is exactly hi + lo, so the same three products apply).  AVEC: 16-B loads allowed
// (K % EPT == 0, aligned).  KFULL: K % 32 == 0.  ZROWS: some rows are padding (graph slots with
// T < 32 / past the batch).
// RBLK: only the first `nblk` of this wavefront's four 32-row blocks hold nodes (f16mx8_core.h): the others' MFMAs are skipped.
template <typename AT, bool AVEC, bool KFULL, bool ZROWS, bool RBLK = false>
__device__ __forceinline__ void mainloop(const AT *const (&arow)[Geom<AT>::NP], const bool (&avalid)[Geom<AT>::NP],
                                         const char *__restrict__ wpack, int K, int k_steps,
                                         int wm, int nt0, int n_tiles_total, char *lds, f32x16 (&acc)[4][RN], int nblk = 4)
{
    using G = Geom<AT>;
    constexpr int EPT = G::EPT, NP = G::NP;
    const int tid = threadIdx.x & (kThreads - 1);   // (a 512-thread workgroup runs two 128-row halves side by side: fused_layer.hip, wide8)
    const int lane = tid & 63;
    const int s_k = (tid % G::TPR) * EPT;  // first k of this thread's 16-B piece

    // Global loads are issued raw (clamped address, no select on the result): the validity select
    // is applied one stage later, at the split -- a select next to the load would make hipcc wait
    // for the load inside the issuing block (measured: 460 -> 640 us).
    float ra[NP][EPT];
    auto load_a_pass = [&](int i, int k0) {
        const int gk = k0 + s_k;
        if constexpr (AVEC) {
            load16<AT>(arow[i] + ((KFULL || gk < K) ? gk : 0), ra[i]);
        } else {
#pragma unroll
            for (int c = 0; c < EPT; ++c) ra[i][c] = elem_to_float(arow[i][(gk + c < K) ? gk + c : 0]);
        }
    };
    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) load_a_pass(i, k0);
    };
    // split pass i (rows of the stage that starts at k0) into the two bf16 planes of buffer `buf`
    auto store_a_pass = [&](int buf, int i, int k0) {
        char *hi_plane = lds + buf * (2 * BM * ROWB);
        char *lo_plane = hi_plane + BM * ROWB;
        const int row = stage_row<AT>(i);
        const int off = a_lds_off(row, s_k >> 3) + (s_k & 4) * 2;
        const int gk = k0 + s_k;
        __bf16 hi[EPT], lo[EPT];
#pragma unroll
        for (int c = 0; c < EPT; ++c) {
            float x = ra[i][c];
            if constexpr (!KFULL || ZROWS) {
                bool in = true;
                if constexpr (!KFULL) in = gk + c < K;
                if constexpr (ZROWS) in = in && avalid[i];
                x = in ? x : 0.0f;
            }
            hi[c] = (__bf16)x;
            lo[c] = (__bf16)(x - (float)hi[c]);
        }
        if constexpr (EPT == 4) {
            *reinterpret_cast<bf16x4 *>(hi_plane + off) = bf16x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<bf16x4 *>(lo_plane + off) = bf16x4{lo[0], lo[1], lo[2], lo[3]};
        } else {
            *reinterpret_cast<bf16x8 *>(hi_plane + off) = bf16x8{hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
            *reinterpret_cast<bf16x8 *>(lo_plane + off) = bf16x8{lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
        }
    };

    // B fragments straight from the packed image; indices clamped, never predicated: a column
    // tile past F duplicates the last real tile and is never stored.
    const char *bbase[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int ntc = nt0 + j < n_tiles_total ? nt0 + j : n_tiles_total - 1;
        bbase[j] = wpack + ((int64_t)ntc * k_steps) * 2 * FRAG_BYTES + lane * 16;
    }
    auto load_b = [&](int ks, bf16x8 (&b)[RN][2]) {  // [col tile][plane]
        ks = ks < k_steps ? ks : k_steps - 1;        // the one-step-ahead prefetch of the last stage
        const int64_t o = (int64_t)ks * 2 * FRAG_BYTES;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            b[j][0] = *reinterpret_cast<const bf16x8 *>(bbase[j] + o);
            b[j][1] = *reinterpret_cast<const bf16x8 *>(bbase[j] + o + FRAG_BYTES);
        }
    };

    const int f_row = wm * 128 + (lane & 31);
    const int f_half = lane >> 5;
    // one 32-row block of one k-step: 2 LDS fragment reads + 6 MFMAs (3 products x 2 column tiles)
    auto mma_block = [&](int buf, int s, int i, const bf16x8 (&b)[RN][2]) {
        if (RBLK && i >= nblk) return;   // wavefront-uniform: a block of padding rows
        // LLVM's MFMA/DS interleave strategy for this scheduling region: measured -4 % on the fused
        // layer (448 vs 467 us, same process, three orderings)
        __builtin_amdgcn_iglp_opt(0);
        const char *hi_plane = lds + buf * (2 * BM * ROWB);
        const char *lo_plane = hi_plane + BM * ROWB;
        const int off = a_lds_off(f_row + i * 32, s * 2 + f_half);
        const bf16x8 a_hi = *reinterpret_cast<const bf16x8 *>(hi_plane + off);
        const bf16x8 a_lo = *reinterpret_cast<const bf16x8 *>(lo_plane + off);
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b[j][0], acc[i][j], 0, 0, 0);
        }
    };

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    static_assert(BK == 32 && NP <= 8, "the interleaved stage below is written for BK = 32");
    bf16x8 b0[RN][2], b1[RN][2];
    const int stages = (K + BK - 1) / BK;
    const int last_k0 = (stages - 1) * BK;

    // ---- prologue: stage 0 split into buffer 0, stage 1's rows already in flight ----
    load_a(0);
    load_b(0, b0);
#pragma unroll
    for (int p = 0; p < NP; ++p) store_a_pass(0, p, 0);
    load_a(BK < last_k0 ? BK : last_k0);
    __syncthreads();

    // ---- one stage = 8 blocks of (2 LDS reads + 6 MFMAs); the split of the NEXT stage's rows
    // (VALU + ds_write) and every global load sit BETWEEN the MFMAs of the same wavefront, so
    // the matrix pipe never waits for a separate "staging phase" (two co-resident workgroups
    // run this loop in lockstep -- measured -- and cannot be relied on to cover one another).
    // A rows are re-requested right after their split (one full stage ahead of the next split),
    // B fragments one k-step ahead; issue order = consumption order (vmcnt retires in order).
    auto stage = [&](int st, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const int k_next1 = (st + 1) * BK < last_k0 ? (st + 1) * BK : last_k0;
        const int k_next2 = (st + 2) * BK;
        const int ka = k_next2 < last_k0 ? k_next2 : last_k0;  // past the end: harmless re-read
        load_b(st * KS + 1, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mma_block(buf, 0, i, b0);
            if (i < NP) {
                store_a_pass(buf ^ 1, i, k_next1);  // rows of stage st+1 -> the other buffer
                load_a_pass(i, ka);         // rows of stage st+2
            }
            if (i + 4 < NP) {
                store_a_pass(buf ^ 1, i + 4, k_next1);
                load_a_pass(i + 4, ka);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        load_b(st * KS + 2, b0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) mma_block(buf, 1, i, b1);
        __syncthreads();  // the only barrier of the stage
    };
    int st = 0;
    for (; st + 1 < stages; st += 2) {
        stage(st, std::integral_constant<int, 0>{});
        stage(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < stages) stage(st, std::integral_constant<int, 0>{});
}

}  // namespace bx3
}  // namespace ggcn
