// LAB ONLY -- a NEGATIVE result kept for the record (round 2): a ping-pong form of the f16mx8 linear.  To rebuild it, copy this
// file into ed-gated-gcn_amd/csrc/ and run `tools/lab.py build pp:` (lab.py compiles every .hip of that directory), then
// `tools/lab.py time linear_pp main@mx8 pp@mx8`.  Measured: bit-identical results, 473 us against 316 us for the plain
// linear at config 2.  A tick (one group's 24-MFMA burst) took 1.6 us instead of the 0.55 us of its matrix time: with one
// register set for the rows in flight the loads are issued only two ticks ahead of their use, so every staging tick waits out
// the HBM latency; a second set does not fit (256 VGPRs, 16 spilled).  One workgroup of 8 wavefronts owns TWO 128 x 256 tiles
// (wavefronts 0-3 / 4-7, the same 256 columns); the two groups run in anti-phase, separated by workgroup
// barriers: while one group issues the 24 MFMAs of a stage, the other splits and stages its next 128 x 32 rows
// and issues its loads.  On every SIMD one wavefront is in a pure matrix burst and its partner in a pure
// VALU / LDS / memory segment (MI355X_MICROARCH.md, "Two waves per SIMD").
//
//   extern "C" int ggcn_lab_linear_pp(X, ldx, wpack, Y, ldy, M, K, F, stream)   (fp32, aligned, K % 32 == 0, M % 256 == 0)
#include "f16mx8_core.h"   // from ed-gated-gcn_amd/csrc when copied there

namespace ggcn {
namespace {

using namespace bx3;
using namespace mx8;

#define GGCN_SB() __builtin_amdgcn_sched_barrier(0)

constexpr int kPPThreads = 512;
constexpr int kGroupLds = 2 * BM * ROWB;    // one stage of one group: fp16 plane + fp8 plane = 16 KiB

__global__ __launch_bounds__(kPPThreads, 2) void linear_pp_kernel(const float *__restrict__ X, int64_t ldx,
                                                                  const char *__restrict__ wpack, float *__restrict__ Y,
                                                                  int64_t ldy, int64_t M, int K, int F, int m_tiles, int n_wg,
                                                                  int stages_packed)
{
    __shared__ __attribute__((aligned(16))) char lds_all[8 * 32 * 64 * 4];   // 64 KiB: 2 x 16 KiB of stage buffers; the store staging reuses all of it
    int m_tile, n_wgi;
    if (!tile_of_block(blockIdx.x, m_tiles, n_wg, m_tile, n_wgi)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int tl = tid & 255;                       // thread index inside the group
    char *lds = lds_all + grp * kGroupLds;
    const int64_t m0 = (int64_t)m_tile * 256 + grp * 128;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wn * RN;
    const int s_k = (tl & 7) * 4;
    const int stages = K / BK;

    const float *arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = X + (m0 + 32 * i + (tl >> 3)) * ldx;

    set_cvt_saturate(true);
    float ra[1][4][4];                              // one stage of rows in flight (two sets spill: 256 VGPRs)
    auto load_a = [&](float (&dst)[4][4], int st) {
        st = st < stages ? st : stages - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = *reinterpret_cast<const float4 *>(arow[i] + st * BK + s_k);
            dst[i][0] = t.x; dst[i][1] = t.y; dst[i][2] = t.z; dst[i][3] = t.w;
        }
    };
    auto stage_rows = [&](const float (&src)[4][4]) {   // split + LDS store of the group's 128 x 32 rows
        char *h_plane = lds, *q_plane = lds + BM * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 32 * i + (tl >> 3);
            const Split4 sp = split4(src[i][0], src[i][1], src[i][2], src[i][3]);
            *reinterpret_cast<uint2 *>(h_plane + a_lds_off(row, s_k >> 3) + (s_k & 4) * 2) = make_uint2(sp.h01, sp.h23);
            const int g = s_k >> 2, hh = (g >> 1) & 1, pos = (g & 1) + 2 * (g >> 2);
            *reinterpret_cast<uint32_t *>(q_plane + q_lds_off(row, hh) + 4 * pos) = (uint32_t)sp.l8;
            *reinterpret_cast<uint32_t *>(q_plane + q_lds_off(row, 2 + hh) + 4 * pos) = (uint32_t)sp.h8;
        }
    };
    const char *bbase[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int ntc = nt0 + j < n_tiles_total ? nt0 + j : n_tiles_total - 1;
        bbase[j] = wpack + (int64_t)ntc * stages_packed * STAGE_PACK_BYTES + lane * 16;
    }
    f16x8 b0[RN], b1[RN];
    i32x4 bq[RN];
    int sq[RN];
    auto load_b = [&](int st) {
        st = st < stages_packed ? st : stages_packed - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const char *p = bbase[j] + (int64_t)st * STAGE_PACK_BYTES;
            b0[j] = *reinterpret_cast<const f16x8 *>(p);
            b1[j] = *reinterpret_cast<const f16x8 *>(p + 1024);
            bq[j] = *reinterpret_cast<const i32x4 *>(p + 2048);
            sq[j] = *reinterpret_cast<const int *>(p + 3072 - lane * 12);
        }
    };
    auto wh8_of = [&](const f16x8 &f0, const f16x8 &f1, int sc) -> i32x4 {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef short s2 __attribute__((ext_vector_type(2)));
        const float inv = __builtin_bit_cast(float, (sc & 0xff00) << 15);
        i32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const f16x8 &f = d < 2 ? f0 : f1;
            const int e = (d & 1) * 4;
            s2 q = __builtin_bit_cast(s2, h2{f[e], f[e + 1]});
            q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, h2{f[e], f[e + 1]}, inv, false);
            q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, h2{f[e + 2], f[e + 3]}, inv, true);
            o[d] = __builtin_bit_cast(int, q);
        }
        return o;
    };
    const int f_row = lane & 31, f_half = lane >> 5;
    const int scale_a = f_half ? SCALE_XH : SCALE_XL;
    auto read_h = [&](int i, f16x8 (&a)[2]) {
        const char *h_plane = lds;
        a[0] = *reinterpret_cast<const f16x8 *>(h_plane + a_lds_off(f_row + 32 * i, f_half));
        a[1] = *reinterpret_cast<const f16x8 *>(h_plane + a_lds_off(f_row + 32 * i, 2 + f_half));
    };
    auto read_q = [&](int i, i32x8 &a) {
        const char *q_plane = lds + BM * ROWB;
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(q_plane + q_lds_off(f_row + 32 * i, f_half));
        const i32x4 hi = *reinterpret_cast<const i32x4 *>(q_plane + q_lds_off(f_row + 32 * i, 2 + f_half));
        a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[4][RN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // the matrix burst of one stage: 16 fp16 + 8 MX MFMAs, fragment reads one block ahead, nothing else
    auto burst = [&]() {
        f16x8 ah[2][2];
        i32x8 aq[2];
        read_h(0, ah[0]);
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) read_h(i + 1, ah[(i + 1) & 1]);
            else read_q(0, aq[0]);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][0], b0[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][0], b0[1], acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][1], b1[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][1], b1[1], acc[i][1], 0, 0, 0);
            GGCN_SB();
        }
        i32x8 bm[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const i32x4 w = wh8_of(b0[j], b1[j], sq[j]);
            bm[j] = i32x8{w[0], w[1], w[2], w[3], bq[j][0], bq[j][1], bq[j][2], bq[j][3]};
        }
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) read_q(i + 1, aq[(i + 1) & 1]);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i & 1], bm[0], acc[i][0], 0, 0, 0, scale_a, 0, sq[0]);
            acc[i][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i & 1], bm[1], acc[i][1], 0, 0, 0, scale_a, 0, sq[1]);
            GGCN_SB();
        }
    };

    // prologue: both groups stage their stage 0; stages 1 and 2 are in flight
    load_a(ra[0], 0);
    load_b(0);
    stage_rows(ra[0]);
    load_a(ra[0], 1);
    __syncthreads();
    // tick t: group (t & 1) bursts its stage t >> 1; the other group stages its next stage (group 1 lags one tick).
    // Group g bursts stage s at tick 2s + g and stages stage s + 1 at tick 2s + 1 + g.
    const int ticks = 2 * stages + 1;
    for (int t = 0; t < ticks; ++t) {
        const int u = t - grp;              // this group's own clock: even = burst, odd = stage
        if (u >= 0 && (u & 1) == 0) {
            if ((u >> 1) < stages) burst();
        } else if (u > 0) {
            const int s1 = (u + 1) >> 1;    // the stage to put into LDS now
            if (s1 < stages) {
                load_b(s1);
                stage_rows(ra[0]);
                load_a(ra[0], s1 + 1);
            }
        }
        __syncthreads();
    }
    set_cvt_saturate(false);

    // store (16-byte row stores through LDS, as linear_split.hip)
    float *stage_lds = reinterpret_cast<float *>(lds_all) + wave * (32 * 64);
    const int c = lane & 31, h = lane >> 5;
    const int colq = (lane & 15) * 4;
    const int gcol = nt0 * NT + colq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row0 = (r & 3) + 8 * (r >> 2);
                stage_lds[(row0 + 4 * h) * 64 + ((32 * j + c) ^ (32 * h))] = acc[i][j][r];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t gm0 = m0 + i * 32;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = 4 * it + (lane >> 4);
            const float4 v4 = *reinterpret_cast<const float4 *>(&stage_lds[row * 64 + (colq ^ (32 * ((row >> 2) & 1)))]);
            if (gcol < F) *reinterpret_cast<float4 *>(Y + (gm0 + row) * ldy + gcol) = v4;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
#undef GGCN_SB

}  // namespace
}  // namespace ggcn

extern "C" int ggcn_lab_linear_pp(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M, int K,
                                  int F, void *stream)
{
    using namespace ggcn;
    if (M % 256 || K % 32 || F % 4 || ldx % 4 || ldy % 4) return fail(GGCN_EUNSUPPORTED, "ggcn_lab_linear_pp: shape");
    const int64_t m_tiles = M / 256;
    const int n_wg = (F + bx3::BN - 1) / bx3::BN;
    const int64_t grid = bx3::grid_for(m_tiles, n_wg);
    hipLaunchKernelGGL(linear_pp_kernel, dim3((unsigned)grid), dim3(kPPThreads), 0, as_stream(stream), X, ldx,
                       static_cast<const char *>(wpack), Y, ldy, M, K, F, (int)m_tiles, n_wg, K / 32);
    return check_launch("ggcn_lab_linear_pp");
}
