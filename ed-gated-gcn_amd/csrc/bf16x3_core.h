// Shared main loop of the bf16x3 matrix kernels (linear_bf16x3.hip, fused_layer.hip).
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
//     (v_cvt_pk_bf16_f32), two bf16 planes in LDS, double-buffered: ONE barrier per 32-deep
//     stage.  LDS rows are 64 B; the 16-B chunk index is XORed with (row>>2)&3 so every
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

namespace ggcn {
namespace bx3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int KSTEP = 16;             // K per MFMA
constexpr int NT = 32;                // columns per MFMA tile
constexpr int FRAG_BYTES = 64 * 16;   // one B fragment: 64 lanes x 8 bf16
constexpr int kThreads = 256;
constexpr int kLdsBytes = 2 * 2 * BM * 64;  // [buffer][plane][128 rows x 64 B] = 32 KiB

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// LDS image of one A plane: [128 rows][4 chunks of 16 B], chunk XOR-swizzled by (row>>2)&3
__device__ __forceinline__ int a_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// XCD-aware remap of a 1-D grid: ids congruent mod 8 share an XCD (observed round-robin
// dispatch; speed only, never correctness); inside one XCD's sequence consecutive ids walk the
// column tiles of the same row block, so they share that block's A rows through the XCD's L2.
__device__ __forceinline__ bool tile_of_block(int id, int m_tiles, int n_wg, int &m_tile, int &n_wgi)
{
    const int xcd = id & 7, slot = id >> 3;
    m_tile = (slot / n_wg) * 8 + xcd;
    n_wgi = slot % n_wg;
    return m_tile < m_tiles;
}
inline int64_t grid_for(int64_t m_tiles, int n_wg) { return (m_tiles + 7) / 8 * 8 * n_wg; }

// arow[i]: this thread's 4 source rows (already clamped to valid memory); avalid[i]: false =>
// the row is padding and must read as zeros.  AVEC: 16-B loads allowed (K % 4 == 0, aligned).
// KFULL: K % 32 == 0.  ZROWS: some rows are padding (graph slots with T < 32 / past the batch).
template <bool AVEC, bool KFULL, bool ZROWS>
__device__ __forceinline__ void mainloop(const float *const (&arow)[4], const bool (&avalid)[4],
                                         const char *__restrict__ wpack, int K, int k_steps,
                                         int nt0, int n_tiles_total, char *lds, f32x16 (&acc)[4][2])
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int s_row = tid >> 3;
    const int s_k4 = (tid & 7) * 4;

    float4 ra[4];
    auto load_a = [&](int k0) {
        const int gk = k0 + s_k4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (AVEC) {
                bool in = true;
                if constexpr (!KFULL) in = gk < K;  // K % 4 == 0: a float4 is all in or all out
                if constexpr (ZROWS) in = in && avalid[i];
                const float4 v = *reinterpret_cast<const float4 *>(arow[i] + ((KFULL || gk < K) ? gk : 0));
                ra[i] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool ink = gk + c < K;
                    const float v = arow[i][ink ? gk + c : 0];
                    e[c] = (ink && (!ZROWS || avalid[i])) ? v : 0.0f;
                }
                ra[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    auto store_a = [&](int buf) {
        char *hi_plane = lds + buf * (2 * BM * 64);
        char *lo_plane = hi_plane + BM * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 32 + s_row;
            const int off = a_lds_off(row, s_k4 >> 3) + (s_k4 & 4) * 2;
            bf16x4 hi, lo;
            hi[0] = (__bf16)ra[i].x; hi[1] = (__bf16)ra[i].y; hi[2] = (__bf16)ra[i].z; hi[3] = (__bf16)ra[i].w;
            lo[0] = (__bf16)(ra[i].x - (float)hi[0]);
            lo[1] = (__bf16)(ra[i].y - (float)hi[1]);
            lo[2] = (__bf16)(ra[i].z - (float)hi[2]);
            lo[3] = (__bf16)(ra[i].w - (float)hi[3]);
            *reinterpret_cast<bf16x4 *>(hi_plane + off) = hi;
            *reinterpret_cast<bf16x4 *>(lo_plane + off) = lo;
        }
    };

    // B fragments straight from the packed image; indices clamped, never predicated: a column
    // tile past F duplicates the last real tile and is never stored.
    const int ntc0 = nt0 < n_tiles_total ? nt0 : n_tiles_total - 1;
    const int ntc1 = nt0 + 1 < n_tiles_total ? nt0 + 1 : n_tiles_total - 1;
    const char *bbase0 = wpack + ((int64_t)ntc0 * k_steps) * 2 * FRAG_BYTES + lane * 16;
    const char *bbase1 = wpack + ((int64_t)ntc1 * k_steps) * 2 * FRAG_BYTES + lane * 16;
    auto load_b = [&](int ks, bf16x8 (&b)[2][2]) {  // [col tile][plane]
        ks = ks < k_steps ? ks : k_steps - 1;       // the one-step-ahead prefetch of the last stage
        const int64_t o = (int64_t)ks * 2 * FRAG_BYTES;
        b[0][0] = *reinterpret_cast<const bf16x8 *>(bbase0 + o);
        b[0][1] = *reinterpret_cast<const bf16x8 *>(bbase0 + o + FRAG_BYTES);
        b[1][0] = *reinterpret_cast<const bf16x8 *>(bbase1 + o);
        b[1][1] = *reinterpret_cast<const bf16x8 *>(bbase1 + o + FRAG_BYTES);
    };

    const int f_row = lane & 31;
    const int f_half = lane >> 5;
    auto mma_step = [&](int buf, int s, const bf16x8 (&b)[2][2]) {
        const char *hi_plane = lds + buf * (2 * BM * 64);
        const char *lo_plane = hi_plane + BM * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = a_lds_off(f_row + i * 32, s * 2 + f_half);
            const bf16x8 a_hi = *reinterpret_cast<const bf16x8 *>(hi_plane + off);
            const bf16x8 a_lo = *reinterpret_cast<const bf16x8 *>(lo_plane + off);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b[j][0], acc[i][j], 0, 0, 0);
            }
        }
    };

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    bf16x8 b0[2][2], b1[2][2];
    const int stages = (K + BK - 1) / BK;
    load_a(0);
    load_b(0, b0);
    for (int st = 0; st < stages; ++st) {
        const int buf = st & 1;
        store_a(buf);
        __syncthreads();  // the only barrier of the stage (double-buffered LDS)
        // issue order = consumption order (vmcnt retires in order): b1 is needed after 24
        // MFMAs, the next A rows only at the next stage's split
        load_b(st * 2 + 1, b1);
        load_a(st + 1 < stages ? (st + 1) * BK : st * BK);  // last stage: harmless re-read
        __builtin_amdgcn_sched_barrier(0);
        mma_step(buf, 0, b0);
        load_b(st * 2 + 2, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(buf, 1, b1);
    }
}

}  // namespace bx3
}  // namespace ggcn
