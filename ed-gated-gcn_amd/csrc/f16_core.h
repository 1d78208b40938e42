// Main loop of the "f16" linear: half-precision FEATURES (BASELINE configs[3]) times the fp16 image of W,
// plain v_mfma_f32_32x32x16_f16 with fp32 accumulation -- 64 matrix-pipe cycles per 32x32x32 block, half of
// f16mx8's, and no operand conversion at all: an fp16 feature IS its own fp16 MFMA operand.
//
// Accuracy: x is exact; w is rounded to fp16 (relative 2^-12 per product, random signs): ~1.4e-4 rms of an O(1)
// output at K = 1024 -- below the 2^-11 rounding of the fp16 OUTPUT itself, and inside config 4's 2e-3 gate
// (the reference cannot run half inputs at all, F7).  fp32 features never take this path.
//
// Geometry as bf16x3_core.h (128 x 256 workgroup tile, 4 wavefronts of 128 x 64, two workgroups per CU), but a
// stage is 64 deep: the two 64-byte "planes" of an LDS row are the two 32-k halves of the stage, so the LDS
// footprint, the chunk swizzle and the fragment reads are the ones of the other two schemes and a barrier is
// paid per 32 MFMAs.  A: one 16-byte global load = 8 consecutive k of a row, stored to LDS as it is
// (ds_write_b128), two stages ahead.  W: the fp16 fragments of the f16mx8 image (the first 2 KiB of every
// 3328-byte record; ggcn_weight_pack(GGCN_PREC_F16MX8)), one 32-k record ahead, straight from L2.
#pragma once
#include "f16mx8_core.h"

namespace ggcn {
namespace f16 {

using namespace bx3;
using mx8::f16x8;

constexpr int BK2 = 64;                       // k per stage
constexpr int NP16 = 4;                       // staging passes: 128 rows x 128 B = 1024 16-byte pieces / 256 threads

#define GGCN_SB() __builtin_amdgcn_sched_barrier(0)

// arow[i]: row pointer of staging pass i (row = 32 i + tid / 8, clamped to valid memory); avalid: padding row -> zeros
template <bool AVEC, bool KFULL, bool ZROWS>
__device__ __forceinline__ void mainloop(const __half *const (&arow)[NP16], const bool (&avalid)[NP16],
                                         const char *__restrict__ wpack, int K, int records, int wm, int nt0,
                                         int n_tiles_total, char *lds, f32x16 (&acc)[4][RN])
{
    static_assert(RN == 2 && WM == 1, "written for 4 wavefronts side by side, 2 column tiles each");
    const int tid = threadIdx.x, lane = tid & 63;
    const int piece = tid & 7;                // 16-byte piece of the row's 128 B: k = 8 piece .. 8 piece + 7
    const int s_row = tid >> 3;               // + 32 per pass
    const int stages = (K + BK2 - 1) / BK2;

    uint4 ra[NP16];
    auto load_a_pass = [&](int i, int st) {
        st = st < stages ? st : stages - 1;
        const int gk = st * BK2 + 8 * piece;
        if constexpr (AVEC) {
            ra[i] = *reinterpret_cast<const uint4 *>(arow[i] + ((KFULL || gk + 8 <= K) ? gk : 0));
        } else {
            union { uint4 v; __half h[8]; } u;
#pragma unroll
            for (int c = 0; c < 8; ++c) u.h[c] = arow[i][(gk + c < K) ? gk + c : 0];
            ra[i] = u.v;
        }
    };
    auto write_pass = [&](int buf, int i, int st) {
        uint4 v = ra[i];
        if constexpr (!KFULL || ZROWS) {
            const int gk = st * BK2 + 8 * piece;
            union { uint4 v; unsigned short h[8]; } u;
            u.v = v;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                bool in = true;
                if constexpr (!KFULL) in = gk + c < K;
                if constexpr (ZROWS) in = in && avalid[i];
                u.h[c] = in ? u.h[c] : (unsigned short)0;
            }
            v = u.v;
        }
        char *plane = lds + buf * (2 * BM * ROWB) + (piece >> 2) * (BM * ROWB);
        *reinterpret_cast<uint4 *>(plane + a_lds_off(32 * i + s_row, piece & 3)) = v;
    };

    // fp16 fragments of record r (32 k) of this wavefront's two column tiles: [k-step 0][k-step 1] at the head of
    // the 3328-byte f16mx8 record
    const char *bbase[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int ntc = nt0 + j < n_tiles_total ? nt0 + j : n_tiles_total - 1;
        bbase[j] = wpack + (int64_t)ntc * records * mx8::STAGE_PACK_BYTES + lane * 16;
    }
    auto load_b = [&](int r, f16x8 (&b)[RN][2]) {
        r = r < records ? r : records - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const char *p = bbase[j] + (int64_t)r * mx8::STAGE_PACK_BYTES;
            b[j][0] = *reinterpret_cast<const f16x8 *>(p);
            b[j][1] = *reinterpret_cast<const f16x8 *>(p + 1024);
        }
    };
    const int f_row = lane & 31, f_half = lane >> 5;
    auto read_a = [&](int buf, int hh, int i, f16x8 (&a)[2]) {
        const char *plane = lds + buf * (2 * BM * ROWB) + hh * (BM * ROWB);
        a[0] = *reinterpret_cast<const f16x8 *>(plane + a_lds_off(f_row + 32 * i, f_half));
        a[1] = *reinterpret_cast<const f16x8 *>(plane + a_lds_off(f_row + 32 * i, 2 + f_half));
    };

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f16x8 b0[RN][2], b1[RN][2];
#pragma unroll
    for (int p = 0; p < NP16; ++p) load_a_pass(p, 0);
    load_b(0, b0);
#pragma unroll
    for (int p = 0; p < NP16; ++p) write_pass(0, p, 0);
#pragma unroll
    for (int p = 0; p < NP16; ++p) load_a_pass(p, 1);
    __syncthreads();

    // One stage = two 32-k halves of 4 row blocks x 4 MFMAs.  Source order = issue order: behind the MFMAs of
    // row block i of the FIRST half go the LDS store of pass i of the next stage and its reload two stages ahead;
    // the fragments of the next row block are read one block ahead of their use, W one record ahead.
    auto stage = [&](int st, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        f16x8 a[2][2];
        read_a(buf, 0, 0, a[0]);
        load_b(2 * st + 1, b1);
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) read_a(buf, 0, i + 1, a[(i + 1) & 1]);
            else read_a(buf, 1, 0, a[0]);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][0], b0[0][0], acc[i][0], 0, 0, 0);
            GGCN_SB();
            write_pass(buf ^ 1, i, st + 1);
            GGCN_SB();
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][0], b0[1][0], acc[i][1], 0, 0, 0);
            GGCN_SB();
            load_a_pass(i, st + 2);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][1], b0[0][1], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][1], b0[1][1], acc[i][1], 0, 0, 0);
            GGCN_SB();
        }
        load_b(2 * st + 2, b0);   // b0 is dead: next stage's first record
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) read_a(buf, 1, i + 1, a[(i + 1) & 1]);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][0], b1[0][0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][0], b1[1][0], acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][1], b1[0][1], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1][1], b1[1][1], acc[i][1], 0, 0, 0);
            GGCN_SB();
        }
        __syncthreads();
    };
    int st = 0;
    for (; st + 1 < stages; st += 2) {
        stage(st, std::integral_constant<int, 0>{});
        stage(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < stages) stage(st, std::integral_constant<int, 0>{});
}
#undef GGCN_SB

}  // namespace f16
}  // namespace ggcn
