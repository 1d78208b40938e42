// Weight gradient dW[K,F] = X[N,K]^T . dH[N,F] (backward of models/gcn.py:34 under train.py:120) as a native "TN" product on
// the bf16 matrix cores: the reduction runs over the NODE rows of both operands, which is the row index they are stored by,
// so both MFMA operands are COLUMNS of a [node][feature] tile.  dweight_bx3.hip brings the product into the forward's shape
// first (a transpose pass of X and a fragment-order pack of dH: two extra round trips of [N,768] through HBM, 40 % of its
// 0.83 ms); here both tiles go to LDS exactly as they lie in memory and gfx950's transposing LDS read (ds_read_b64_tr_b16,
// cdna guide T10) delivers their columns as operand fragments:
//
//   per stage = 16 nodes = ONE k-step of v_mfma_f32_32x32x16_bf16:
//     X  tile [16 nodes][128 features]  fp32 -> bf16 hi + lo planes (x = hi + lo, residual 2^-17 |x|)      8 KiB
//     dH tile [16 nodes][256 features]  likewise, as two 128-column images                                  16 KiB
//     LDS image of a plane: 256-byte rows, 16-byte chunk ch of row r at ch ^ (((r&3)<<2) | ((r>>2)&3))  (T10 image (b):
//     conflict-free for the 8-byte writes of the staging pass and for the transposed reads)
//     A fragment of output rows 32 i .. 32 i + 31 (= X features): lane (r, h) holds nodes 8h .. 8h+7 of column 32 i + r;
//     B fragment of output columns likewise from the dH image -- two ds_read_b64_tr_b16 each
//     acc[i][j] += Ahi.Bhi + Alo.Bhi + Ahi.Blo          (bf16x3: the fp32 exponent range gradients need, ~1e-5 relative)
//   workgroup = 4 wavefronts = a [128 x 256] tile of dW over one chunk of the node axis (split-K); wavefront = 128 x 64;
//   double-buffered LDS (48 KiB), global loads one stage ahead in registers, one barrier per stage; two workgroups per CU.
//   The chunks' partial tiles go to a [S][K][F] slab array and are added in a fixed order (no float atomics: bitwise
//   reproducible), as in dweight_bx3.hip, whose plan (chunks per XCD) this kernel shares.
// Needs 16-byte aligned rows (ldx, ldg multiples of 4); other shapes keep the transpose form.
#include "bf16x3_core.h"
#include "common.h"

#include <cstdlib>

namespace ggcn {
namespace {

using namespace bx3;

constexpr int TN_NODES = 16;                 // nodes per stage
constexpr int TN_BM = 128, TN_BN = 256;      // output tile: X features x dH features
constexpr int kPlaneX = TN_NODES * 256;      // 4 KiB: [16][128] bf16
constexpr int kTnBuf = 2 * kPlaneX + 4 * kPlaneX;   // X hi, lo + dH (2 images x hi, lo) = 24 KiB
constexpr int kTnLds = 2 * kTnBuf;           // 48 KiB

__device__ __forceinline__ int tn_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

typedef short s16x4 __attribute__((ext_vector_type(4)));

// 4 fp32 -> bf16 hi and lo quadruples (lo = bf16(v - hi)); vectors, so that hipcc packs pairs with v_cvt_pk_bf16_f32
__device__ __forceinline__ void split_bf16x4(const float4 &v, bf16x4 &hi, bf16x4 &lo)
{
    const float x[4] = {v.x, v.y, v.z, v.w};
    __bf16 h[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        h[c] = (__bf16)x[c];
        l[c] = (__bf16)(x[c] - (float)h[c]);
    }
    hi = bf16x4{h[0], h[1], h[2], h[3]};
    lo = bf16x4{l[0], l[1], l[2], l[3]};
}

// One transposed read of 4 x 16-bit per lane; OFF: a compile-time byte offset (the lo plane lies kPlaneX behind the hi plane)
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned addr)
{
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// byte offset, inside a [16][128] plane image, of this lane's share of the transposed read that delivers nodes
// 8h + 4 half .. + 3 of column 32 blk + (lane & 31)  (T10: lane 4q + p of a 16-lane group supplies row q, columns 4p .. 4p + 3)
__device__ __forceinline__ int tr_lane_off(int blk, int half, int lane)
{
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int h = g >> 1;
    const int c0 = blk * 32 + 16 * (g & 1);
    return tn_off(8 * h + 4 * half + q, (c0 >> 3) + (p >> 1)) + 8 * (p & 1);
}
// hi and lo fragments (8 values each) of one column block from two lane addresses (the two row halves)
__device__ __forceinline__ void tr_fragments(unsigned a0, unsigned a1, bf16x8 &hi, bf16x8 &lo)
{
    union { bf16x8 v; s16x4 half[2]; } uh, ul;
    uh.half[0] = tr_read<0>(a0); uh.half[1] = tr_read<0>(a1);
    ul.half[0] = tr_read<kPlaneX>(a0); ul.half[1] = tr_read<kPlaneX>(a1);
    hi = uh.v; lo = ul.v;
}

// FULLC: K % 128 == 0 and F % 256 == 0 (no column of a tile lies outside the matrices)
template <bool FULLC>
__global__ __launch_bounds__(kThreads, 2) void dweight_tn_kernel(const float *__restrict__ X, int64_t ldx,
                                                                 const float *__restrict__ G, int64_t ldg, int64_t N, int K, int F,
                                                                 float *__restrict__ slabs, int64_t chunk_rows, int n_splits,
                                                                 int m_tiles, int n_wg)
{
    __shared__ __attribute__((aligned(16))) char lds[kTnLds];
    // Work item w = split * tiles + tile; block ids that share an XCD (id & 7: observed dispatch, speed only) take a CONTIGUOUS
    // run of ceil(total / 8) items, so that the tiles of one chunk of rows sit on one XCD -- two at most, split between its
    // X-feature blocks (tile = m_tile * n_wg + n_wgi) -- and share that chunk's rows through its L2.  Round 4 gave every XCD
    // whole chunks only: 3 x 18 = 54 of its 64 resident slots at config 2's shape (432 workgroups on 512 slots); now 28 chunks
    // x 18 tiles = 504.
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int tiles = m_tiles * n_wg;
    const int total = n_splits * tiles, per_xcd = (total + 7) >> 3;
    const int w = xcd * per_xcd + qb;
    if (qb >= per_xcd || w >= total) return;  // whole workgroup, before any barrier
    const int split = w / tiles, tile = w - split * tiles;
    const int m_tile = tile / n_wg, n_wgi = tile % n_wg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = m_tile * TN_BM, n0 = n_wgi * TN_BN;
    const int64_t row0 = (int64_t)split * chunk_rows;
    const int64_t rows_end = row0 + chunk_rows < N ? row0 + chunk_rows : N;
    const int stages = (int)((chunk_rows + TN_NODES - 1) / TN_NODES);

    // staging: X tile = 512 pieces of 16 B (2 per thread), dH tile = 1024 pieces (4 per thread); piece idx: node = idx / (cols/4)
    int xnode[2], xcol[2], gnode[4], gcol[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) { const int idx = tid + 256 * p; xnode[p] = idx >> 5; xcol[p] = (idx & 31) * 4; }
#pragma unroll
    for (int p = 0; p < 4; ++p) { const int idx = tid + 256 * p; gnode[p] = idx >> 6; gcol[p] = (idx & 63) * 4; }
    float4 rx[2], rg[4];
    // Both operands behind buffer resources whose extent ends with the chunk's last row: a stage past the chunk (or the tail
    // of the last chunk) reads zeros by itself, the lane offsets are fixed before the loop and the stage offset is a scalar --
    // no 64-bit lane arithmetic, no validity test per stage.  (The launcher keeps a chunk's rows below 2 GiB.)
    constexpr int kRsrcFlags = 0x00020000;   // raw buffer, 32-bit elements (gfx9 family)
    const int64_t rows_here = rows_end > row0 ? rows_end - row0 : 0;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + row0 * ldx), 0, (int)(rows_here * ldx * 4), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G + row0 * ldg), 0, (int)(rows_here * ldg * 4), kRsrcFlags);
    uint32_t xoff[2], goff[4];
    bool xin[2], gin[4];     // this piece's columns exist (whole pieces: K % 4 == F % 4 == 0)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        xin[p] = FULLC || m0 + xcol[p] < K;
        xoff[p] = (uint32_t)(((int64_t)xnode[p] * ldx + (xin[p] ? m0 + xcol[p] : 0)) * 4);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        gin[p] = FULLC || n0 + gcol[p] < F;
        goff[p] = (uint32_t)(((int64_t)gnode[p] * ldg + (gin[p] ? n0 + gcol[p] : 0)) * 4);
    }
    const uint32_t xstep = (uint32_t)(TN_NODES * ldx * 4), gstep = (uint32_t)(TN_NODES * ldg * 4);
    auto load_x = [&](int s, int p) { rx[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[p], (uint32_t)s * xstep, 0)); };
    auto load_g = [&](int s, int p) { rg[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gr, goff[p], (uint32_t)s * gstep, 0)); };
    auto load_stage = [&](int s) {
#pragma unroll
        for (int p = 0; p < 2; ++p) load_x(s, p);
#pragma unroll
        for (int p = 0; p < 4; ++p) load_g(s, p);
    };
    int xo[2], go[4];   // LDS offsets of this thread's pieces inside a buffer
#pragma unroll
    for (int p = 0; p < 2; ++p) xo[p] = tn_off(xnode[p], xcol[p] >> 3) + (xcol[p] & 4) * 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) go[p] = 2 * kPlaneX + (gcol[p] >> 7) * 2 * kPlaneX + tn_off(gnode[p], (gcol[p] & 127) >> 3) + (gcol[p] & 4) * 2;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto write_x = [&](int s, int p) {   // register piece p of stage s -> planes of buffer s & 1
        char *b = lds + (s & 1) * kTnBuf;
        bf16x4 hi, lo;
        split_bf16x4(FULLC || xin[p] ? rx[p] : zero4, hi, lo);
        *reinterpret_cast<bf16x4 *>(b + xo[p]) = hi;
        *reinterpret_cast<bf16x4 *>(b + kPlaneX + xo[p]) = lo;
    };
    auto write_g = [&](int s, int p) {
        char *b = lds + (s & 1) * kTnBuf;
        bf16x4 hi, lo;
        split_bf16x4(FULLC || gin[p] ? rg[p] : zero4, hi, lo);
        *reinterpret_cast<bf16x4 *>(b + go[p]) = hi;
        *reinterpret_cast<bf16x4 *>(b + kPlaneX + go[p]) = lo;
    };
    auto write_stage = [&](int s) {
#pragma unroll
        for (int p = 0; p < 2; ++p) write_x(s, p);
#pragma unroll
        for (int p = 0; p < 4; ++p) write_g(s, p);
    };

    f32x16 acc[4][RN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // lane offsets of the transposed reads inside a buffer: A = X plane image, column block i; B = this wavefront's two column
    // tiles (columns 64 wave + 32 j of the 256: image (64 wave + 32 j) >> 7, block ((64 wave + 32 j) & 127) >> 5)
    int aoff[4][2], boff[RN][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) aoff[i][hf] = tr_lane_off(i, hf, lane);
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int col = 64 * wave + 32 * j;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) boff[j][hf] = 2 * kPlaneX + (col >> 7) * 2 * kPlaneX + tr_lane_off((col & 127) >> 5, hf, lane);
    }
    load_stage(0);
    write_stage(0);
    load_stage(1);   // (past the chunk: zeros)
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        const char *b = lds + (s & 1) * kTnBuf;
        bf16x8 bh[RN], bl[RN], ah[2], al[2];
        const unsigned lb = (unsigned)(size_t)b;
#pragma unroll
        for (int j = 0; j < RN; ++j) tr_fragments(lb + boff[j][0], lb + boff[j][1], bh[j], bl[j]);
        tr_fragments(lb + aoff[0][0], lb + aoff[0][1], ah[0], al[0]);
        // The transposed reads are asm: the compiler does not count them, and nothing but a data dependency keeps an MFMA
        // behind the wait -- so every fragment passes through an (empty) asm that follows its wait.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < RN; ++j) { asm volatile("" : "+v"(bh[j])); asm volatile("" : "+v"(bl[j])); }
        asm volatile("" : "+v"(ah[0]));
        asm volatile("" : "+v"(al[0]));
        // The source order below IS the issue order (a sched_barrier after every group, as in the forward loops): each row
        // block's six MFMAs carry a share of the next stage's split + LDS writes (the other buffer: last read a stage ago,
        // behind the barrier below; unconditional -- past the chunk the pieces are zeros nobody reads) and, once a piece's
        // registers are free, its global load two stages ahead.
#define GGCN_SBT() __builtin_amdgcn_sched_barrier(0)
        GGCN_SBT();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3)     // the next row block's fragments under this block's MFMAs
                tr_fragments(lb + aoff[i + 1][0], lb + aoff[i + 1][1], ah[(i + 1) & 1], al[(i + 1) & 1]);
            GGCN_SBT();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i & 1], bh[0], acc[i][0], 0, 0, 0);   // small terms first
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bl[0], acc[i][0], 0, 0, 0);
            GGCN_SBT();
            if (i == 0) { write_x(s + 1, 0); write_x(s + 1, 1); }
            if (i == 1) { write_g(s + 1, 0); write_g(s + 1, 1); }
            if (i == 2) { write_g(s + 1, 2); write_g(s + 1, 3); }
            GGCN_SBT();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bh[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i & 1], bh[1], acc[i][1], 0, 0, 0);
            GGCN_SBT();
            if (i == 0) { load_x(s + 2, 0); load_x(s + 2, 1); }   // (past the chunk: zeros from the buffer bounds, unused)
            if (i == 1) { load_g(s + 2, 0); load_g(s + 2, 1); }
            if (i == 2) { load_g(s + 2, 2); load_g(s + 2, 3); }
            GGCN_SBT();
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bl[1], acc[i][1], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bh[1], acc[i][1], 0, 0, 0);
            GGCN_SBT();
            if (i < 3) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(ah[(i + 1) & 1]));
                asm volatile("" : "+v"(al[(i + 1) & 1]));
            }
        }
#undef GGCN_SBT
        __syncthreads();
    }

    float *slab = slabs + (int64_t)split * K * F;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int gn = n0 + 64 * wave + 32 * j + (lane & 31);
        if (gn >= F) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gmb = m0 + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = gmb + (r & 3) + 8 * (r >> 2);
                if (gm < K) slab[(int64_t)gm * F + gn] = acc[i][j][r];
            }
        }
    }
}

// ---- the 256 x 256 tile: one workgroup of four wavefronts per CU, each with a 128 x 128 share = 256 accumulator registers (AGPRs)
// of its 512 ----------------------------------------------------------------------------------------------------------
// dweight_tn_kernel runs at the board's power cap (1400 W: its time is its energy), and a large term of that energy is what it
// stages: per node row the 768 X features are brought into an LDS by the 3 column-group tiles and the 768 gradient values by the 6
// row-group tiles -- 3.6 GB per launch at config 2.  A 256 x 256 tile stages [16][256] of both operands per 16 nodes for twice the
// output area: a third less.  Same plane images, same transposed reads, the same products in the same order per output element
// and the same chunks of the node axis as the 128 x 256 kernel: bit-identical results.
constexpr int T2_B = 256;
constexpr int kT2Buf = 8 * kPlaneX;          // X: 2 images x (hi, lo), dH: 2 images x (hi, lo) = 32 KiB
constexpr int kT2Lds = 2 * kT2Buf;           // 64 KiB

__global__ __launch_bounds__(kThreads, 1) void dweight_tn256_kernel(const float *__restrict__ X, int64_t ldx,
                                                                    const float *__restrict__ G, int64_t ldg, int64_t N, int K, int F,
                                                                    float *__restrict__ slabs, int64_t chunk_rows, int n_splits,
                                                                    int m_tiles, int n_wg)
{
    __shared__ __attribute__((aligned(16))) char lds[kT2Lds];
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int tiles = m_tiles * n_wg;
    const int total = n_splits * tiles, per_xcd = (total + 7) >> 3;
    const int w = xcd * per_xcd + qb;
    if (qb >= per_xcd || w >= total) return;  // whole workgroup, before any barrier
    const int split = w / tiles, tile = w - split * tiles;
    const int m_tile = tile / n_wg, n_wgi = tile % n_wg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = m_tile * T2_B, n0 = n_wgi * T2_B;
    const int64_t row0 = (int64_t)split * chunk_rows;
    const int64_t rows_end = row0 + chunk_rows < N ? row0 + chunk_rows : N;
    const int stages = (int)((chunk_rows + TN_NODES - 1) / TN_NODES);

    // staging: both tiles are [16 nodes][256 columns] fp32 = 1024 pieces of 16 B, 4 per thread: node = idx / 64, column = 4 (idx % 64)
    int pnode[4], pcol[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { const int idx = tid + 256 * p; pnode[p] = idx >> 6; pcol[p] = (idx & 63) * 4; }
    float4 rx[4], rg[4];
    constexpr int kRsrcFlags = 0x00020000;
    const int64_t rows_here = rows_end > row0 ? rows_end - row0 : 0;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + row0 * ldx), 0, (int)(rows_here * ldx * 4), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G + row0 * ldg), 0, (int)(rows_here * ldg * 4), kRsrcFlags);
    uint32_t xoff[4], goff[4];
    int po[4];     // LDS offset of piece p inside its operand's half of a buffer
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        xoff[p] = (uint32_t)(((int64_t)pnode[p] * ldx + m0 + pcol[p]) * 4);
        goff[p] = (uint32_t)(((int64_t)pnode[p] * ldg + n0 + pcol[p]) * 4);
        po[p] = (pcol[p] >> 7) * 2 * kPlaneX + tn_off(pnode[p], (pcol[p] & 127) >> 3) + (pcol[p] & 4) * 2;
    }
    const uint32_t xstep = (uint32_t)(TN_NODES * ldx * 4), gstep = (uint32_t)(TN_NODES * ldg * 4);
    auto load_x = [&](int s, int p) { rx[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[p], (uint32_t)s * xstep, 0)); };
    auto load_g = [&](int s, int p) { rg[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gr, goff[p], (uint32_t)s * gstep, 0)); };
    auto write_x = [&](int s, int p) {
        char *b = lds + (s & 1) * kT2Buf;
        bf16x4 hi, lo;
        split_bf16x4(rx[p], hi, lo);
        *reinterpret_cast<bf16x4 *>(b + po[p]) = hi;
        *reinterpret_cast<bf16x4 *>(b + kPlaneX + po[p]) = lo;
    };
    auto write_g = [&](int s, int p) {
        char *b = lds + (s & 1) * kT2Buf + 4 * kPlaneX;
        bf16x4 hi, lo;
        split_bf16x4(rg[p], hi, lo);
        *reinterpret_cast<bf16x4 *>(b + po[p]) = hi;
        *reinterpret_cast<bf16x4 *>(b + kPlaneX + po[p]) = lo;
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // A = the X image of this wavefront's 128 features (image wr), block i; B = the dH image wc, block j
    int aoff[4][2], boff[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            aoff[i][hf] = wr * 2 * kPlaneX + tr_lane_off(i, hf, lane);
            boff[i][hf] = 4 * kPlaneX + wc * 2 * kPlaneX + tr_lane_off(i, hf, lane);
        }
#pragma unroll
    for (int p = 0; p < 4; ++p) { load_x(0, p); load_g(0, p); }
#pragma unroll
    for (int p = 0; p < 4; ++p) { write_x(0, p); write_g(0, p); }
#pragma unroll
    for (int p = 0; p < 4; ++p) { load_x(1, p); load_g(1, p); }   // (past the chunk: zeros)
    __syncthreads();
#define GGCN_SBT() __builtin_amdgcn_sched_barrier(0)
    for (int s = 0; s < stages; ++s) {
        const unsigned lb = (unsigned)(size_t)(lds + (s & 1) * kT2Buf);
        bf16x8 bh[4], bl[4], ah[2], al[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_fragments(lb + boff[j][0], lb + boff[j][1], bh[j], bl[j]);
        tr_fragments(lb + aoff[0][0], lb + aoff[0][1], ah[0], al[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(bh[j])); asm volatile("" : "+v"(bl[j])); }
        asm volatile("" : "+v"(ah[0]));
        asm volatile("" : "+v"(al[0]));
        GGCN_SBT();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) tr_fragments(lb + aoff[i + 1][0], lb + aoff[i + 1][1], ah[(i + 1) & 1], al[(i + 1) & 1]);
            GGCN_SBT();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i & 1], bh[j], acc[i][j], 0, 0, 0);   // small terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bl[j], acc[i][j], 0, 0, 0);
                GGCN_SBT();
                // a share of the next stage's split + LDS writes (the other buffer) and of the loads two stages ahead per MFMA group
                if (j == 0) write_x(s + 1, i);
                if (j == 1) write_g(s + 1, i);
                GGCN_SBT();
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bh[j], acc[i][j], 0, 0, 0);
                GGCN_SBT();
                if (j == 2) load_x(s + 2, i);
                if (j == 3) load_g(s + 2, i);
                GGCN_SBT();
            }
            if (i < 3) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(ah[(i + 1) & 1]));
                asm volatile("" : "+v"(al[(i + 1) & 1]));
            }
        }
        __syncthreads();
    }
#undef GGCN_SBT
    float *slab = slabs + (int64_t)split * K * F;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gn = n0 + 128 * wc + 32 * j + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gmb = m0 + 128 * wr + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(int64_t)(gmb + (r & 3) + 8 * (r >> 2)) * F + gn] = acc[i][j][r];
        }
    }
}

__global__ __launch_bounds__(256) void tn_slab_sum_kernel(const float *__restrict__ slabs, int n_slabs, int64_t kf, int F,
                                                         float *__restrict__ dW, int64_t lddw)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= kf) return;
    float s = 0.0f;
    for (int z = 0; z < n_slabs; ++z) s += slabs[(int64_t)z * kf + i];  // fixed order
    dW[(i / F) * lddw + (i % F)] = s;
}

struct TnPlan { int m_tiles, n_wg, n_splits; int64_t chunk_rows; size_t slab_bytes; bool big; };

// The 256 x 256 tile where it pays: whole tiles (K, F multiples of 256) and enough node rows that one workgroup per CU still gets
// chunks of a thousand rows and more (131 072 x 768 x 768: 0.600 J against 0.638 J per launch, each alone at the 1400 W cap: 429 vs
// 456 us; the training step 2.89-2.91 vs 2.93-2.98 ms; 20 000 rows: 98 vs 96 us, 5 000 x 256 x 512: 49 vs 36 us -- those keep the
// 128 x 256 tile).  A pure function of the shape: ggcn_dweight_workspace_bytes and ggcn_dweight can never disagree.
bool tn_big_tile(int64_t N, int K, int F)
{
    bool big = K % T2_B == 0 && F % T2_B == 0 && N >= 65536;
#ifdef GGCN_LAB_DW   // lab builds only (tools/labbuild.sh "-DGGCN_LAB_DW"): GGCN_DW_TILE=128 / 256 picks the tile per call
    if (const char *e = getenv("GGCN_DW_TILE")) big = K % T2_B == 0 && F % T2_B == 0 && atoi(e) == 256;
#endif
    return big;
}

TnPlan tn_plan(int64_t N, int K, int F)
{
    TnPlan p;
    p.big = tn_big_tile(N, K, F);
    p.m_tiles = p.big ? K / T2_B : (K + TN_BM - 1) / TN_BM;
    p.n_wg = p.big ? F / T2_B : (F + TN_BN - 1) / TN_BN;
    const int tiles = p.m_tiles * p.n_wg;
    int64_t s = (p.big ? 256 : 512) / tiles;   // as many chunks as fill the chip's resident slots (two 128 x 256 workgroups per CU, or one 256 x 256) in ONE round
    if (s < 8) s = 8;                  // (more tiles than slots: eight chunks, several rounds)
    const int64_t max_s = (N + 511) / 512;   // at least 512 node rows per chunk
    if (s > max_s) s = max_s;
#ifdef GGCN_LAB_DW   // lab builds only (tools/dw_timing.py through tools/labbuild.sh "-DGGCN_LAB_DW"): the product plan is a pure
                     // function of (N, K, F), so ggcn_dweight_workspace_bytes and ggcn_dweight can never disagree about the slabs
    if (const char *e = getenv("GGCN_LAB_DW_SPLITS")) s = atoi(e);
#endif
    if (s < 1) s = 1;
    p.n_splits = (int)s;
    const int64_t rows = (N + s - 1) / s;
    p.chunk_rows = (rows + TN_NODES - 1) / TN_NODES * TN_NODES;
    p.slab_bytes = (size_t)p.n_splits * K * F * sizeof(float);
    return p;
}

}  // namespace

bool dweight_tn_takes(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F)
{
    if (!((ldx % 4 == 0) && (ldg % 4 == 0) && (K % 4 == 0) && (F % 4 == 0) && aligned16(X) && aligned16(G))) return false;
    if (N <= 0 || K <= 0 || F <= 0) return false;
    // a chunk's rows sit behind one buffer resource (2 GiB): wider rows keep the transpose form (dweight_bx3.hip)
    const TnPlan p = tn_plan(N, K, F);
    return (p.chunk_rows + TN_NODES) * (ldx > ldg ? ldx : ldg) * 4 < ((int64_t)1 << 31);
}

size_t dweight_tn_workspace_bytes(int64_t N, int K, int F)
{
    if (N <= 0 || K <= 0 || F <= 0) return 0;
    return (tn_plan(N, K, F).slab_bytes + 255) & ~(size_t)255;
}

int dweight_tn(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F, float *dW, int64_t lddw,
               void *workspace, hipStream_t st)
{
    const TnPlan p = tn_plan(N, K, F);
    float *slabs = static_cast<float *>(workspace);
    const int tiles = p.m_tiles * p.n_wg;
    const int64_t grid = (int64_t)8 * (((int64_t)p.n_splits * tiles + 7) / 8);   // 8 XCDs x ceil(total / 8) work items
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_dweight: too many tiles");
    if ((p.chunk_rows + TN_NODES) * (ldx > ldg ? ldx : ldg) * 4 >= ((int64_t)1 << 31))
        return fail(GGCN_EUNSUPPORTED, "ggcn_dweight: a chunk of %lld rows of %lld floats exceeds the 2 GiB a buffer resource addresses",
                    (long long)p.chunk_rows, (long long)(ldx > ldg ? ldx : ldg));
    if (p.big)
        hipLaunchKernelGGL(dweight_tn256_kernel, dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, G, ldg, N, K, F, slabs,
                           p.chunk_rows, p.n_splits, p.m_tiles, p.n_wg);
    else if (K % TN_BM == 0 && F % TN_BN == 0)
        hipLaunchKernelGGL(dweight_tn_kernel<true>, dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, G, ldg, N, K, F, slabs,
                           p.chunk_rows, p.n_splits, p.m_tiles, p.n_wg);
    else
        hipLaunchKernelGGL(dweight_tn_kernel<false>, dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, G, ldg, N, K, F, slabs,
                           p.chunk_rows, p.n_splits, p.m_tiles, p.n_wg);
    const int64_t kf = (int64_t)K * F;
    hipLaunchKernelGGL(tn_slab_sum_kernel, dim3((unsigned)((kf + 255) / 256)), dim3(256), 0, st, slabs, p.n_splits, kf, F, dW, lddw);
    return check_launch("ggcn_dweight(bf16x3, TN)");
}

}  // namespace ggcn
